# round-2 experiment A: NF / ruf-readback variants, single-sweep timings
for v in base nf4 rg rgnf4; do
  cp pam_amd/lib$v.so pam_amd/libpam_amd_awfl.so
  bash tools/exp_ab.sh "$v"
done
cp pam_amd/librg.so pam_amd/libpam_amd_awfl.so
for fl in 0 32768 49152; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --lds-floor $fl --chunks 0 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('rg floor=$fl', round(d['value']/1e9,4), 'G/s', round(d['ms_per_step'],1), 'ms', {k:round(v['avg_ms'],3) for k,v in d['kernels'].items() if k in ('flux','update','fct_mult')})"
done
cp pam_amd/libbase.so pam_amd/libpam_amd_awfl.so
for m in 1 2 4; do
  PAMA_SWEEP_MASK=$m timeout -k 10 200 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --chunks 1 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('base mask=$m', round(d['ms_per_step'],1), 'ms', {k:round(v['avg_ms'],3) for k,v in d['kernels'].items() if k in ('flux','update','fct_mult')})"
done
