mkdir -p gpurun_out/r4
python bench.py --steps 40 --warmup 5 > gpurun_out/r4/bench_f64f.json 2> gpurun_out/r4/bench_f64f.err
python -m pytest tests -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r4/t_full1.txt
cat gpurun_out/r4/t_full1.txt
