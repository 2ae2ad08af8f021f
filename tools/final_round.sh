#!/bin/bash
# End-of-round GPU run, part A: the whole GPU suite, smoke(), and the bench lines kept under profiles/r03_*.  (Part B = tools/profile_round.sh.)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
O=gpurun_out/r03_final
mkdir -p $O
timeout -k 10 900 python -m pytest tests -q -m gpu -x > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -3 $O/smoke.log
timeout -k 10 600 python bench.py > $O/bench_line.json 2> $O/bench_line.err; echo default line done
F1="--gop 0 --frames-in-flight 1 --no-cpu-baseline"
timeout -k 10 300 python bench.py $F1 --with-upload > $O/bench_line_f1.json 2>/dev/null; echo f1 done
timeout -k 10 300 python bench.py --width 3840 --height 2160 --no-cpu-baseline > $O/bench_line_4k.json 2>/dev/null; echo 4k done
timeout -k 10 300 python bench.py $F1 --width 3840 --height 2160 --inter-frac 0 > $O/bench_line_4k_all_intra_f1.json 2>/dev/null; echo 4k all-intra f1 done
timeout -k 10 300 python bench.py --gop 0 --frames-in-flight 8 --no-cpu-baseline --width 3840 --height 2160 --inter-frac 0 --no-verify > $O/bench_line_4k_all_intra_f8.json 2>/dev/null; echo 4k all-intra f8 done
timeout -k 10 400 python bench.py $F1 --inter-frac 0 > $O/bench_line_8k_all_intra_f1.json 2>/dev/null; echo 8k all-intra f1 done
timeout -k 10 300 python bench.py --width 1920 --height 1080 --bd 8 --no-cpu-baseline --no-verify > $O/bench_line_1080p8.json 2>/dev/null; echo 1080p done
echo all done
