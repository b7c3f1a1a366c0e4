cd /root/repo
Q="--no-cpu-baseline --no-mrr --batch-sweep= --traffic off --standalone-steps 0 --cold-items 0"
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 $Q > /dev/null 2>&1; done
for rep in 1 2 3; do for u in 2 1; do
SBR_SCORE_U=$u python bench.py --steps 20 --warmup 5 $Q 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('U=$u', round(d['ms_per_step'],4), 'score us', round(r['avg_launch_ms']*1e3,1), round(r['frac'],3))"
done; done
