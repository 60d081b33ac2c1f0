mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf gpurun_out/prof_v5
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_v5 -o v5 --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/prof_v5.log 2>&1 || exit 1
tail -1 gpurun_out/prof_v5.log | cut -c1-120
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_v5.log 2>&1 || exit 1
tail -1 gpurun_out/bench_v5.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --classes 1000 > gpurun_out/c1000_v5.log 2>&1 || exit 1
tail -1 gpurun_out/c1000_v5.log | cut -c1-260
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --arch vit_l14_336 --batch 128 --classes 1000 > gpurun_out/vitl_v5.log 2>&1 || exit 1
tail -1 gpurun_out/vitl_v5.log | cut -c1-260
python tools/cocoop_bench.py 2>&1 | tail -1
