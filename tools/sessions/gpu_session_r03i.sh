# GPU-box session r03i: per-kernel step profile of BOTH operand builds on one box (where do the fp16 build's extra ~4 % go?)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
for dt in bf16 fp16; do
  rocprofv3 --kernel-trace --stats -d $o/prof_$dt -o step -- python bench.py --dtype $dt --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-roofline --no-torch-baseline --no-other-dtype > $o/prof_${dt}_step.log 2>&1
  python tools/export_profile.py $o/prof_$dt/step_results.db $o/r03i_step_kernel_stats_$dt.csv 3
  rm -rf $o/prof_$dt
done
python - <<'PY'
import csv, re
def load(f):
    d={}
    for r in csv.DictReader(open(f)):
        k=re.sub(r'\(.*','',r['kernel']).replace('void ','')[:44]+' g'+r['grid_threads_x']
        d[k]=d.get(k,0)+float(r['total_ms_per_step'])
    return d
a,b=load('gpurun_out/r03i_step_kernel_stats_bf16.csv'),load('gpurun_out/r03i_step_kernel_stats_fp16.csv')
print('total bf16 %.1f fp16 %.1f'%(sum(a.values()),sum(b.values())))
for k in sorted(set(a)|set(b), key=lambda k:-max(a.get(k,0),b.get(k,0)))[:28]:
    print('%-58s %8.2f %8.2f %+6.2f'%(k,a.get(k,0),b.get(k,0),b.get(k,0)-a.get(k,0)))
PY
