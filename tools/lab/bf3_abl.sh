# per-kernel times of the bf16-prefilter retrieval under lab knobs (MF_BF3_ABL) / seed sample strides (MF_BF3_BS); run through gpurun.
# The knobs exist only in the lab build:  make -C matrix-factorization-torch_amd/csrc BUILD=_build_lab LIB=../lib/libmf_hip_lab.so EXTRA=-DMF_BF3_LAB
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
export MF_HIP_LIB=$GRAFT_REPO_ROOT/matrix-factorization-torch_amd/lib/libmf_hip_lab.so
run() {
tag=$1; shift
export "$@"
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/lab/prof_$tag -o p -- python3 tools/lab/bf3_probe.py 1024 > gpurun_out/lab/prof_$tag.log 2>&1
f=$(ls gpurun_out/lab/prof_$tag/*kernel_stats.csv gpurun_out/lab/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
echo "== $tag"; grep "path bf16" gpurun_out/lab/prof_$tag.log
python3 - "$f" <<PY
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    if "bf3" in r["Name"] and "build" not in r["Name"]: print("  ", r["Name"][:60].ljust(60), r["Calls"], r["AverageNs"])
PY
rm -rf gpurun_out/lab/prof_$tag
unset MF_BF3_BS MF_BF3_ABL MF_BF3_XT
}
for t in "$@"; do case $t in base) run base MF_X=0;; bs*) run $t MF_BF3_BS=${t#bs};; abl*) run $t MF_BF3_ABL=${t#abl};; xt2) run xt2 MF_BF3_XT=2;; xt2abl*) run $t MF_BF3_XT=2 MF_BF3_ABL=${t#xt2abl};; esac; done
