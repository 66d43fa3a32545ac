set -u
ROOT=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python3 $ROOT/benchmarks/pu_me_bench.py --cpu-pus 0 --reps 1"
mkdir -p $ROOT/gpurun_out/pu_pmc
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $ROOT/gpurun_out/pu_pmc/sq -- $B > /dev/null 2>&1 && echo sq ok
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $ROOT/gpurun_out/pu_pmc/ta -- $B > /dev/null 2>&1 && echo ta ok
python3 $ROOT/profiles/summarize_pmc.py $ROOT/gpurun_out/pu_pmc > $ROOT/gpurun_out/pu_pmc_summary.json
