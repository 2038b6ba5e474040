mkdir -p gpurun_out
for st in 1 2 3; do
  timeout -k 10 300 python bench.py --steps 6 --warmup 2 --streams $st --no-cpu-baseline > gpurun_out/bench_streams$st.json 2> gpurun_out/bench_streams$st.err || { tail -5 gpurun_out/bench_streams$st.err; exit 1; }
  python -c "
import json; d=json.load(open('gpurun_out/bench_streams$st.json')); print('streams', $st, 'value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2), 'iters', d['config']['outer_iterations_max'], 'trace', d['config']['trace_estimate'])"
done
