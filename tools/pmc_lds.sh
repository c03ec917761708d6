# PMC passes on the single-stream pipeline (GPU box): where do K4 wave cycles go?
set -e
export TMPDIR=/tmp
O=gpurun_out/pmc_lds
mkdir -p $O
timeout -k 10 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/a -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-sweep --streams 1 > $O/a.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/b -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-sweep --streams 1 > $O/b.log 2>&1
