# GPU-box session r03h: GEMM variants (TN on 16-row MFMAs, one phase everywhere, no setprio) same-box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
o=gpurun_out
for lib in default gemm_tn16 gemm_ph16all gemm_noprio default; do
  if [ $lib = default ]; then unset PXA_LIB_PATH; else export PXA_LIB_PATH=pixart_sigma_amd/variants/lib_$lib.so; fi
  timeout 300 python tools/kbench.py gemm 2>&1 | grep -v amdgpu.ids | grep -v "split_k=[24]"
done > $o/r03h_kbench_gemm_variants.txt
unset PXA_LIB_PATH
python - <<'PY'
import re
rows=[]; cur=None
for l in open('gpurun_out/r03h_kbench_gemm_variants.txt'):
    if l.startswith('lib:'): cur=l.split()[-1].split('/')[-1]; rows.append((cur,{})); continue
    m=re.match(r'gemm (\w+) (\w+).*?([\d.]+) TF/s',l)
    if m: rows[-1][1][m.group(1)+' '+m.group(2)]=float(m.group(3))
print('%-10s '%''+' '.join('%14s'%r[0][-14:] for r in rows))
for k in rows[0][1]: print('%-10s '%k+' '.join('%14.1f'%r[1].get(k,0) for r in rows))
PY
