set -o pipefail
timeout -k 10 500 python bench.py > gpurun_out/fin_bench_1024.json 2> gpurun_out/fin_bench_1024.err || exit 1
timeout -k 10 300 python bench.py --size 2048 --optimizer lbfgs --precision bf16 --steps 10 --warmup 3 > gpurun_out/fin_bench_bf16.json 2> gpurun_out/fin_bench_bf16.err || exit 1
python -c "
import json
for f in ('fin_bench_1024','fin_bench_bf16'):
    d=json.load(open('gpurun_out/%s.json'%f)); print(f, round(d['value'],2), d['timing']['ms_per_step'], d['roofline']['frac'], d.get('worker_level'), d.get('parity',{}).get('loss_rel'), d.get('cpu_baseline',{}).get('value'))"
bash tools/profile_round.sh zf > gpurun_out/prof_zf.log 2>&1 || exit 1
tail -2 gpurun_out/prof_zf.log
