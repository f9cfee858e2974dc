# the mined step's prefilter scan under lab knobs (MF_MBF_ABL: 1 = hits not stored, 2 = no compares, 4 = no MFMAs, 8 = no barriers); run through gpurun.
# The knobs exist only in the lab build:  make -C matrix-factorization-torch_amd/csrc BUILD=_build_lab LIB=../lib/libmf_hip_lab.so EXTRA=-DMF_BF3_LAB
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/lab
export MF_HIP_LIB=$GRAFT_REPO_ROOT/matrix-factorization-torch_amd/lib/libmf_hip_lab.so
for a in "$@"; do
  export MF_MBF_ABL=$a
  echo "== MF_MBF_ABL=$a"
  timeout -k 10 200 python3 tools/lab/mined_timeline.py 2>&1 | grep "span mining\|prefilter on" | tail -6
done
