#!/bin/bash
# scripts/visit.sh -- ONE parametrised GPU visit (replaces the ~150 one-off scripts/gpu_r*_*.sh of rounds 1-4).
#   gpurun --timeout T -- 'VISIT=r5a STAGES="tests smoke bench prof pmc extra fuzz layout" bash scripts/visit.sh'
# Stages (any subset, run in this order; results under gpurun_out/$VISIT):
#   tests   the whole `pytest -m gpu` suite WITHOUT -x (TESTS="path::id ..." narrows it)       -> pytest_gpu.log
#   smoke   __graft_entry__.smoke()
#   bench   python bench.py $BENCH_ARGS (default --steps 50 --warmup 5)                         -> bench_n1.json
#   prof    rocprofv3 --kernel-trace --stats of bench.py (and of the rne leg at 1e7)            -> prof/, prof_rne1e7/
#   profsec rocprofv3 --kernel-trace --stats of the FULL bench.py command (secondary legs included); per (kernel, grid) durations    -> secondary_kernel_stats.csv
#   pmc     FETCH_SIZE / WRITE_SIZE passes (separate runs) of bench.py and of the rne leg       -> pmc_*/
#   sq      SQ_INSTS_VALU / SQ_WAVES of bench_extra.py --what $SQ_WHAT (default rne,dyn,tree)        -> pmc_sq/, sq_digest.txt
#   extra   python bench_extra.py $EXTRA_ARGS                                                   -> bench_extra.jsonl
#   fuzz    scripts/gpu_fuzz_*.py (random robots through every kernel size against the oracle)  -> fuzz_*.jsonl
#   layout  scripts/layout_probe.py --fleet (packed vs two-array outputs over fresh allocations; VARIANTS="a.so b.so" adds A/B libraries)
#   ikab    scripts/ik_ab.py over VARIANTS (A/B libraries of the IK kernel, interleaved, sustained)
#   libab   scripts/ik_lib_time.py for the product library and every VARIANTS library, LIBAB_ROUNDS interleaved rounds         -> ik_lib_ab.jsonl
#   iksq    scripts/ik_loss_factors.py plainly and under an SQ_INSTS_VALU / SQ_WAVES pass                                      -> ik_loss_factors.json, ik_sq.json
#   place   scripts/placement_pmc.py: allocator variants of one 464 MB non-temporal stream, TCC counter passes                  -> placement.jsonl, place_table*.txt
#   cmd     eval "$CMD" (anything else; output -> cmd.log)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd $(dirname $0)/.. && pwd)}
O=$R/gpurun_out/${VISIT:-visit}
mkdir -p $O
cd $R
has() { [[ " $STAGES " == *" $1 "* ]]; }
if has tests; then
  timeout ${TESTS_TIMEOUT:-1500} python -m pytest ${TESTS:-tests} -m gpu -q -rf --timeout 900 > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/pytest_gpu.log
  grep -E "passed|failed|FAILED|ERROR|rc=" $O/pytest_gpu.log | tail -15
fi
if has smoke; then timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1; fi
if has bench; then
  timeout 600 python bench.py ${BENCH_ARGS:---steps 50 --warmup 5} > $O/bench_n1.json 2> $O/bench_n1.err; cut -c1-300 $O/bench_n1.json; tail -3 $O/bench_n1.err
  python scripts/bench_digest.py $O/bench_n1.json
fi
if has layout; then
  timeout 400 python scripts/layout_probe.py --fleet --tag product > $O/layout_probe.jsonl 2> $O/layout_probe.err; tail -2 $O/layout_probe.err
  for v in $VARIANTS; do
    RTBHIP_LIB=$R/robotics-toolbox-python_amd/lib/variants/$v timeout 300 python scripts/layout_probe.py --tag $v >> $O/layout_probe.jsonl 2>> $O/layout_probe.err
  done
  grep -v summary $O/layout_probe.jsonl | cut -c1-330
fi
if has prof; then
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o bench -- python $R/bench.py --steps 40 --warmup 3 --no-cpu --no-secondary > $O/prof.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_rne1e7 -o rne -- python $R/bench_extra.py --what rne --no-cpu --steps 40 --n-rne 10000000 > $O/prof_rne1e7.log 2>&1
  cd $R
  find $O/prof $O/prof_rne1e7 -name "*kernel_stats*.csv" | while read f; do echo $f; cut -c1-170 "$f" | head -6; done
fi
if has profsec; then
  cd /tmp
  timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_secondary -o sec -- python $R/bench.py --steps 20 --warmup 3 --no-cpu > $O/prof_secondary.log 2>&1
  cd $R
  python scripts/secondary_stats.py $O/prof_secondary $O/secondary_kernel_stats.csv | cut -c1-170
  tail -1 $O/prof_secondary.log | cut -c1-300
fi
if has pmc; then
  cd /tmp
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 5 --warmup 2 --no-cpu --no-secondary > $O/pmc_$c.log 2>&1 || echo "pmc $c failed"
    timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_rne_$c -o pmc -- python $R/bench_extra.py --what rne --no-cpu --steps 5 > $O/pmc_rne_$c.log 2>&1 || echo "pmc rne $c failed"
  done
  cd $R
  python scripts/pmc_digest.py $O
fi
if has sq; then
  # VALU instructions per wave of the issue-bound kernels (bench_extra.py VALU_PER_UNIT): SQ_INSTS_VALU / SQ_WAVES, its own pass
  cd /tmp
  timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_sq -o sq -- python $R/bench_extra.py --what ${SQ_WHAT:-rne,dyn,tree} --no-cpu --steps 3 > $O/pmc_sq.log 2>&1 || echo "sq pass failed"
  cd $R
  python scripts/pmc_digest.py $O | grep pmc_sq > $O/sq_digest.txt; cut -c1-200 $O/sq_digest.txt | head -60
fi
if has extra; then
  timeout 900 python bench_extra.py $EXTRA_ARGS > $O/bench_extra.jsonl 2> $O/bench_extra.err; cut -c1-200 $O/bench_extra.jsonl; tail -2 $O/bench_extra.err
fi
if has ikab; then
  timeout ${IKAB_TIMEOUT:-900} python scripts/ik_ab.py $IKAB_ARGS > $O/ik_ab.jsonl 2> $O/ik_ab.err; cut -c1-260 $O/ik_ab.jsonl | tail -60; tail -3 $O/ik_ab.err
fi
if has libab; then
  # A/B LIBRARIES of k_ik (robotics-toolbox-python_amd/lib/variants/*.so from scripts/build_ik_variant.sh), one process per library and round, interleaved
  : > $O/ik_lib_ab.jsonl
  for r in $(seq 1 ${LIBAB_ROUNDS:-3}); do
    for v in product $VARIANTS; do
      L=""; T=""                                    # an entry with '=' is an rtbhip_tune setting of the product library, else a variant library;
      if [[ $v == *:* ]]; then L=$R/robotics-toolbox-python_amd/lib/variants/${v%%:*}; T=${v#*:};      # "lib.so:key=value": both
      elif [[ $v == *=* ]]; then T=$v; elif [ $v != product ]; then L=$R/robotics-toolbox-python_amd/lib/variants/$v; fi
      RTBHIP_LIB=$L RTBHIP_TUNE=$T timeout 200 python scripts/ik_lib_time.py >> $O/ik_lib_ab.jsonl 2>> $O/ik_lib_ab.err
    done
  done
  python scripts/ik_lib_digest.py $O/ik_lib_ab.jsonl
fi
if has fuzz; then
  for f in ${FUZZ:-dyn ik kin rne paths fleet jit}; do
    timeout 900 python scripts/gpu_fuzz_$f.py > $O/fuzz_$f.jsonl 2> $O/fuzz_$f.err; echo "fuzz $f rc=$?" >> $O/fuzz_$f.jsonl; tail -2 $O/fuzz_$f.jsonl | cut -c1-300
  done
fi
if has iksq; then
  # k_ik's executed VALU instructions per wave iteration: SQ_INSTS_VALU / SQ_WAVES of config 3's call (its own pass), divided by the wave iterations
  # the same script reports from the kernel's diagnostic counters -> ik_sq.json (committed as profiles/r06_ik_sq.json: the bench line's constant)
  timeout 300 python scripts/ik_loss_factors.py > $O/ik_loss_factors.json 2> $O/ik_loss_factors.err; cut -c1-600 $O/ik_loss_factors.json
  cd /tmp
  mkdir -p $O/pmc_iksq
  timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $O/pmc_iksq -o sq -- python $R/scripts/ik_loss_factors.py --launches 4 > $O/pmc_iksq/run.json 2> $O/pmc_iksq.log || echo "ik sq pass failed"
  cd $R
  python scripts/ik_loss_factors.py --digest $O/pmc_iksq $O/ik_sq.json | cut -c1-500
fi
if has place; then
  # placement of ONE non-temporal 464 MB output stream (scripts/placement_pmc.py): allocator variants timed plainly, then counter passes on torch's arrays
  : > $O/placement.jsonl
  for al in ${PLACE_ALLOCS:-torch hip ext:0x3 ext:0x4 align:2 align:1024}; do
    timeout 300 python scripts/placement_pmc.py --alloc $al >> $O/placement.jsonl 2>> $O/placement.err
  done
  grep summary $O/placement.jsonl | cut -c1-300
  (rocprofv3 -L 2>/dev/null || rocprofv3 --list-avail 2>/dev/null) | grep -o "TCC_[A-Z0-9_a-z]*\|MALL[A-Z0-9_a-z]*" | sort -u > $O/tcc_counters.txt; wc -l $O/tcc_counters.txt
  cd /tmp
  i=0
  while read -r set; do
    [ -z "$set" ] && continue
    i=$((i+1))
    mkdir -p $O/pmc_place$i
    timeout 400 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/pmc_place$i -o pl -- python $R/scripts/placement_pmc.py --launches 12 > $O/pmc_place$i/run.jsonl 2> $O/pmc_place$i.log || echo "place pass $i ($set) failed"
    python $R/scripts/placement_pmc.py --digest $O/pmc_place$i $O/pmc_place$i/run.jsonl > $O/place_table$i.txt 2>> $O/placement.err; cat $O/place_table$i.txt | cut -c1-250
  done <<< "${PLACE_SETS:-TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum
TCC_TAG_STALL_sum TCC_BUSY_sum
TCC_HIT_sum TCC_MISS_sum TCC_WRITEBACK_sum
TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_GMI_sum TCC_EA0_WRREQ_IO_sum}"
  cd $R
fi
if has cmd; then eval "$CMD" > $O/cmd.log 2>&1; tail -${CMD_TAIL:-40} $O/cmd.log; fi
find $O -name "*kernel_trace.csv" -size +8M -delete
find $O -name "*counter_collection.csv" -size +8M -delete
exit 0
