#!/bin/bash
# The ONE parameterised GPU-box runner (replaces the per-experiment gpu_job_*.sh of rounds 1-4; profiles/INDEX.md maps every kept
# summary to the task that produced it).
#
#   gpurun --timeout 900 -- 'bash scripts/gpu_job.sh <tag> <task> [<task> ...]'
#
# Every task writes into gpurun_out/ under the names profiles/ uses (<tag>_<what>), so a summary is committed by copying it.
# Tasks (a task argument follows after a colon, e.g. pytest:tests/test_fp8_gpu.py; pytest / bench / py arguments are eval'ed, so a -k expression
# with spaces is written pytest:"tests/x.py -k 'a or b'"):
#   pytest[:<pytest args>]   python -m pytest <args or `tests -m gpu -x -q`>            -> <tag>_pytest.txt
#   bench[:<bench args>]     python bench.py <args or `--steps 20 --warmup 5`>          -> <tag>_bench.json
#   smoke                    __graft_entry__.smoke()
#   headline                 rocprofv3 --kernel-trace --stats of 3 headline steps       -> <tag>_kernel_stats.md
#   b1 | b32                 the same at batch 1 / 32 (cold steps)                      -> <tag>_b1_kernel_stats.md
#   warm | inc               12 warm / incremental env steps (scripts/warm_steps.py)    -> <tag>_warm_kernel_stats.md
#   warm1 | inc1             the same at batch 1 (the reference eval loop's steady state)
#   gpt | gato | flamingo    kernel trace of a baseline policy's cold steps at batch 256     -> <tag>_<policy>_kernel_stats.md
#   pmc                      FETCH_SIZE / WRITE_SIZE passes + per-shape join            -> <tag>_pmc_traffic.{md,json}
#   sq                       SQ counter pass                                            -> <tag>_sq_counters.md
#   py:<script args>         python <script args> (stdout -> <tag>_py_<n>.txt)
#   sh:<command>             bash -c <command>    (stdout -> <tag>_sh_<n>.txt)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out; mkdir -p $O
TAG=${1:?tag}; shift
export TMPDIR=/tmp
n=0
prof() {   # prof <name> <title> <command...>: kernel trace of a command, summarised
  local name=$1 title=$2; shift 2
  rm -rf /tmp/prof_$name
  (cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_$name -o t -- "$@" > $O/${TAG}_${name}_stdout.txt 2>&1)
  python $R/scripts/rocprof_summary.py /tmp/prof_$name/t_results.db $O/${TAG}_${name}_kernel_stats.md "$title" | tail -1
  tail -1 $O/${TAG}_${name}_stdout.txt | cut -c1-200
}
for task in "$@"; do
  arg=""; case "$task" in *:*) arg="${task#*:}"; task="${task%%:*}";; esac
  n=$((n+1))
  cd $R
  case "$task" in
    pytest) eval "timeout 1500 python -m pytest ${arg:-tests -m gpu -x -q}" > $O/${TAG}_pytest_$n.txt 2>&1; tail -4 $O/${TAG}_pytest_$n.txt;;
    bench) S=$(date +%s); eval "timeout 900 python bench.py ${arg:---steps 20 --warmup 5}" > $O/${TAG}_bench_$n.json 2> $O/${TAG}_bench_${n}_stderr.txt
           echo "bench.py ${arg:---steps 20 --warmup 5}: $(( $(date +%s) - S )) s wall"; cut -c1-300 $O/${TAG}_bench_$n.json; tail -3 $O/${TAG}_bench_${n}_stderr.txt;;
    smoke) python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3;;
    headline) prof headline "$TAG: rocprofv3 --kernel-trace --stats -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --opt dual_stream=0 $arg (VIMA-200M B=256 Lp=512 bf16, 1x MI355X)" \
                python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --headline-only --opt dual_stream=0 $arg
              mv $O/${TAG}_headline_kernel_stats.md $O/${TAG}_kernel_stats.md;;
    b1|b32) prof $task "$TAG: batch ${task#b} COLD steps, dual_stream=0, 8 steps (rocprofv3 --kernel-trace --stats -- python bench.py --batch ${task#b} --steps 6 --warmup 2 --no-cpu-baseline --headline-only --opt dual_stream=0 $arg)" \
                python $R/bench.py --batch ${task#b} --steps 6 --warmup 2 --no-cpu-baseline --headline-only --opt dual_stream=0 $arg;;
    warm|inc) prof $task "$TAG: 12 ${task} steps (prompt K/V cached: obs ViT + decoder + action head) B=256, VIMA-200M Lp=512 bf16 + one prompt assembly (python scripts/warm_steps.py $task 10 256 $arg)" \
                python $R/scripts/warm_steps.py $task 10 256 $arg;;
    warm1|inc1) prof $task "$TAG: ${task%1} steps at BATCH 1 (the reference eval loop's steady state: one obs ViT + decoder step + action head), VIMA-200M Lp=512 bf16 + one prompt assembly (python scripts/warm_steps.py ${task%1} 40 1 $arg)" \
                python $R/scripts/warm_steps.py ${task%1} 40 1 $arg;;
    gpt|gato|flamingo) prof $task "$TAG: $task BASELINE policy, batch 256, cold steps, dual_stream=0 (rocprofv3 --kernel-trace --stats -- python bench.py --policy $task --batch 256 --steps 4 --warmup 2 --opt dual_stream=0 $arg)" \
                python $R/bench.py --policy $task --batch 256 --steps 4 --warmup 2 --opt dual_stream=0 $arg;;
    pmc) for c in FETCH_SIZE WRITE_SIZE; do
           rm -rf /tmp/pmc_$c
           (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_$c -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --headline-only --live-pmc off --opt dual_stream=0 --launch-log /tmp/launches.json $arg > /dev/null 2>&1)
         done
         python $R/scripts/pmc_summary.py /tmp/pmc_FETCH_SIZE/bench_results.db /tmp/pmc_WRITE_SIZE/bench_results.db $O/${TAG}_pmc_traffic.md $O/${TAG}_pmc_traffic.json /tmp/launches.json | tail -2;;
    sq) rm -rf /tmp/pmc_sq
        (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE -d /tmp/pmc_sq -o bench -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --headline-only --live-pmc off --opt dual_stream=0 $arg > /dev/null 2>&1)
        python $R/scripts/sq_summary.py /tmp/pmc_sq/bench_results.db $O/${TAG}_sq_counters.md | tail -1;;
    py) eval "timeout 900 python $arg" > $O/${TAG}_py_$n.txt 2>&1; tail -12 $O/${TAG}_py_$n.txt | cut -c1-300;;
    sh) timeout 900 bash -c "$arg" > $O/${TAG}_sh_$n.txt 2>&1; tail -12 $O/${TAG}_sh_$n.txt | cut -c1-300;;
    *) echo "gpu_job.sh: unknown task $task";;
  esac
done
