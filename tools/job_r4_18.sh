mkdir -p /root/repo/gpurun_out/r4; rm -f gpurun_out/parity_measured.jsonl
python -m pytest tests -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r4/t_full2.txt
cat gpurun_out/r4/t_full2.txt
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r4/smoke.txt 2>&1; tail -2 gpurun_out/r4/smoke.txt
python bench.py > gpurun_out/r4/bench_full.json 2> gpurun_out/r4/bench_full.err; tail -c 300 gpurun_out/r4/bench_full.json
