#!/bin/bash
# rocprofv3 passes for the bench workload (run on the GPU box through gpurun): kernel trace + stats,
# HBM bytes (FETCH_SIZE and WRITE_SIZE in separate passes, as MI355X_MICROARCH.md prescribes), SQ
# issue counters; then the same for K1b on config C5.  Outputs land in gpurun_out/prof/; the
# summaries to be judged are copied into profiles/ afterwards (scripts/collect_profiles.py).
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/prof
rm -rf $O && mkdir -p $O
ARGS="--steps ${STEPS:-10} --warmup 3 --cpu-seconds 0 --highs-seconds 0 --tto-seconds 0 --others 0 --no-dive-leg 0 ${EXTRA:-}"
echo "$ARGS" > $O/args.txt
# the kernel trace over the bench's own default timed region (200 steps), so that its average launch time
# is the figure the default bench.py run measures live; the counter passes take fewer steps
TARGS="--cpu-seconds 0 --highs-seconds 0 --tto-seconds 0 --others 0 --no-dive-leg 0 ${EXTRA:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py $TARGS > $O/trace.log 2>&1
echo trace done
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python3 $R/bench.py $ARGS > $O/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python3 $R/bench.py $ARGS > $O/write.log 2>&1
echo hbm done
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $O/sq1 -- python3 $R/bench.py $ARGS > $O/sq1.log 2>&1 || echo "sq1 failed"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $O/sq2 -- python3 $R/bench.py $ARGS > $O/sq2.log 2>&1 || echo "sq2 failed"
echo sq done
# K1b on C5 (1024 x 512): trace + HBM bytes
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k1b_trace -- python3 $R/scripts/c5_tree.py 1024 8 1 > $O/k1b_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/k1b_fetch -- python3 $R/scripts/c5_tree.py 1024 8 1 > $O/k1b_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/k1b_write -- python3 $R/scripts/c5_tree.py 1024 8 1 > $O/k1b_write.log 2>&1
echo k1b done
# C4 (cut rounds in the engine): kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4_trace -- python3 $R/scripts/c4_tree.py 256 128 4096 10 > $O/c4_trace.log 2>&1
echo c4 done
# K1c (one cold 1024 x 512 LP over the chip, one launch per pivot): kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/k1c_trace -- python3 $R/scripts/root_coop.py 1024 512 > $O/k1c_trace.log 2>&1
echo k1c done
python3 $R/scripts/collect_profiles.py $O
