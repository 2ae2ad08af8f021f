set -e
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
timeout -k 10 1000 python -m pytest tests -q -m gpu -x > gpurun_out/r02_final_tests.log 2>&1; tail -3 gpurun_out/r02_final_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r02_final_smoke.log 2>&1; tail -3 gpurun_out/r02_final_smoke.log
bash tools/profile_round.sh > gpurun_out/r02_final_profile.log 2>&1; tail -2 gpurun_out/r02_final_profile.log
cd $R
timeout -k 10 500 python bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench_line.err
timeout -k 10 300 python bench.py --frames-in-flight 1 --no-cpu-baseline --with-upload > gpurun_out/r02_bench_line_f1.json 2>/dev/null
timeout -k 10 300 python bench.py --frames-in-flight 4 --no-cpu-baseline --no-verify > gpurun_out/r02_bench_line_f4.json 2>/dev/null
timeout -k 10 300 python bench.py --width 3840 --height 2160 --no-cpu-baseline --no-verify > gpurun_out/r02_bench_line_4k.json 2>/dev/null
timeout -k 10 300 python bench.py --width 1920 --height 1080 --bd 8 --no-cpu-baseline --no-verify > gpurun_out/r02_bench_line_1080p8.json 2>/dev/null
echo all done
