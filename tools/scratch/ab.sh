mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/t_gpu.log 2>&1 || { tail -30 gpurun_out/t_gpu.log; exit 1; }
tail -2 gpurun_out/t_gpu.log
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/trim_new.log 2>&1 || exit 1
MUDPT_LIB=/root/repo/mudpt_amd/lib/libmudpt_ref.so python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/trim_ref.log 2>&1 || exit 1
python - <<PY
import json
for v in ('new','ref'):
    d=json.loads(open(f'gpurun_out/trim_{v}.log').read().strip().splitlines()[-1])
    print(v, d['ms_per_step'], d['value'], d['roofline']['achieved'], d['config']['final_loss'])
PY
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --classes 1000 > gpurun_out/trim_c1000.log 2>&1 && tail -1 gpurun_out/trim_c1000.log | cut -c1-330
MUDPT_LIB=/root/repo/mudpt_amd/lib/libmudpt_ref.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --classes 1000 > gpurun_out/trim_c1000_ref.log 2>&1 && tail -1 gpurun_out/trim_c1000_ref.log | cut -c1-330
python tools/cocoop_bench.py > gpurun_out/trim_cocoop.log 2>&1; tail -3 gpurun_out/trim_cocoop.log
