#!/bin/bash
# Round profiles on the GPU box (run through gpurun from the repository root):
#   bash tools/collect_profiles.sh r04
# kernel trace + stats of the default bench.py run, then three PMC passes (SQ counters, FETCH_SIZE, WRITE_SIZE; counters in
# their own runs, no trace domains beside them), summarised into gpurun_out/profiles_<round>/ -- copy into profiles/.
set -o pipefail
R=${1:-r04}
OUT=gpurun_out/profiles_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python3 bench.py > $OUT/bench_stdout.log 2>&1 || exit 1
grep '^{"metric"' $OUT/bench_stdout.log | tail -1 > $OUT/${R}_bench.json
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$R -o bench -- python3 bench.py > $OUT/prof_stdout.log 2>&1 || exit 1
cp $(ls gpurun_out/prof_$R/*kernel_stats.csv gpurun_out/prof_$R/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${R}_bench_kernel_stats.csv
grep '^{"metric"' $OUT/prof_stdout.log | tail -1 > $OUT/${R}_bench_under_rocprof.json
rm -rf gpurun_out/prof_$R
# the headline workload alone (no extras: the B = 32 / d = 64 legs launch the same kernel names at other sizes), whose
# per-kernel averages are the ones to hold against the roofline objects' avg_ms
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$R -o bench -- python3 bench.py --no-extras > $OUT/prof_headline_stdout.log 2>&1 || exit 1
cp $(ls gpurun_out/prof_$R/*kernel_stats.csv gpurun_out/prof_$R/*/*kernel_stats.csv 2>/dev/null | head -1) $OUT/${R}_bench_headline_kernel_stats.csv
grep '^{"metric"' $OUT/prof_headline_stdout.log | tail -1 > $OUT/${R}_bench_headline_under_rocprof.json
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \
    -d gpurun_out/pmcS_$R -o r -- python3 bench.py --steps 3 --warmup 1 --no-extras > $OUT/pmcS_stdout.log 2>&1 || exit 1
python3 tools/pmc_sq.py $(ls gpurun_out/pmcS_$R/*.db gpurun_out/pmcS_$R/*/*.db 2>/dev/null | head -1) > $OUT/${R}_pmc_sq_summary.json || exit 1
timeout -k 10 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmcF_$R -o r -- python3 bench.py --steps 3 --warmup 1 --no-extras > $OUT/pmcF_stdout.log 2>&1 || exit 1
timeout -k 10 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmcW_$R -o r -- python3 bench.py --steps 3 --warmup 1 --no-extras > $OUT/pmcW_stdout.log 2>&1 || exit 1
python3 tools/pmc_traffic.py $(ls gpurun_out/pmcF_$R/*.db gpurun_out/pmcF_$R/*/*.db 2>/dev/null | head -1) $(ls gpurun_out/pmcW_$R/*.db gpurun_out/pmcW_$R/*/*.db 2>/dev/null | head -1) > $OUT/${R}_traffic.json || exit 1
# kernel timeline of one steady training step (rocpd database of a short headline run)
timeout -k 10 600 rocprofv3 --kernel-trace -d gpurun_out/tl_$R -o tl -- python3 bench.py --no-extras --no-cpu-baseline > $OUT/tl_stdout.log 2>&1 || exit 1
python3 tools/step_timeline.py $(ls gpurun_out/tl_$R/*.db gpurun_out/tl_$R/*/*.db 2>/dev/null | head -1) > $OUT/${R}_step_timeline.txt || exit 1
rm -rf gpurun_out/tl_$R
# the reference's default loss (4 mined negatives): kernel timeline of one steady step, and the SQ counters of its kernels
timeout -k 10 300 rocprofv3 --kernel-trace -d gpurun_out/tlm_$R -o m -- python3 tools/lab/mined_timeline.py > $OUT/tlm_stdout.log 2>&1 || exit 1
python3 tools/lab/mined_timeline.py $(ls gpurun_out/tlm_$R/*.db gpurun_out/tlm_$R/*/*.db 2>/dev/null | head -1) > $OUT/${R}_mined_step_timeline.txt || exit 1
grep "mined step" $OUT/tlm_stdout.log | sed 's/^/# (same run, under rocprofv3) /' >> $OUT/${R}_mined_step_timeline.txt
rm -rf gpurun_out/tlm_$R
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS \
    -d gpurun_out/pmcM_$R -o r -- python3 tools/lab/mined_timeline.py > $OUT/pmcM_stdout.log 2>&1 || exit 1
python3 tools/pmc_sq.py $(ls gpurun_out/pmcM_$R/*.db gpurun_out/pmcM_$R/*/*.db 2>/dev/null | head -1) > $OUT/${R}_mined_pmc_sq.json || exit 1
rm -rf gpurun_out/pmcM_$R
# the raw rocprof databases are far beyond what travels back
rm -rf gpurun_out/prof_$R gpurun_out/pmcS_$R gpurun_out/pmcF_$R gpurun_out/pmcW_$R
ls -la $OUT
