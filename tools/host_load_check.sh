# graph replay vs eager launches with the host CPUs saturated by busy loops (what a shared box does to the enqueue path)
n=${1:-320}
for i in $(seq $n); do timeout 100 python3 -c "while True: pass" & done
sleep 2
for a in "" "--no-graph" "" "--no-graph"; do
  python bench.py --steps 60 --warmup 10 --no-cpu-baseline $a 2>/dev/null | python -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0]); print(round(d['ms_per_step'],3), 'median', round(d.get('step_ms_median', 0),3), d['config'].get('launch',''))"
done
wait
