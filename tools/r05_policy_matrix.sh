# HEAD against the round-start build over other shapes (sbr_rs_amd/libsbr_hip_old.so built from commit 6574761 by hand)
cd /root/repo
Q="--no-cpu-baseline --no-mrr --batch-sweep= --traffic off --standalone-steps 0 --cold-items 0"
for i in 1 2 3; do python bench.py --steps 20 --warmup 3 $Q > /dev/null 2>&1; done
one() { # label lib args...
  label=$1; lib=$2; shift; shift
  SBR_HIP_LIB=$lib python bench.py --steps 12 --warmup 3 $Q "$@" 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k={n: round(v['ms_per_launch'],3) for n,v in d.get('kernels',{}).items()}
print('$label', round(d['value']/1e6,2), 'M/s', round(d['ms_per_step'],4), 'ms', k)"
}
NEW=/root/repo/sbr_rs_amd/libsbr_hip.so; OLD=/root/repo/sbr_rs_amd/libsbr_hip_old.so
while read -r name args; do
  for rep in 1 2; do
    one "$name old" $OLD $args
    one "$name always" /root/repo/sbr_rs_amd/libsbr_hip_always.so $args
    one "$name new" $NEW $args
  done
done <<'CFG'
b8192 --batch-sequences 8192
b12288 --batch-sequences 12288
b16384 --batch-sequences 16384
b32768 --batch-sequences 32768
b50000 --batch-sequences 50000
ewma128_warp --model ewma --loss warp
coupled_warp --model lstm-coupled
CFG
