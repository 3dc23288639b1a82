# GPU-box session r04b: which kernel does hipBLASLt run for the NT shapes where it is ahead (name encodes tile / MFMA / waves / prefetch), with its registers and LDS
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for shape in "65536 4608 1152" "65536 1152 4608" "65536 3456 1152" "65536 1152 1152"; do
  KBENCH_LIBREF=1 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_lt -o lt -- python tools/kbench_one.py $shape NT 10 > /dev/null 2>&1
  python - "$shape" <<'PY' >> gpurun_out/r04b_hipblaslt_kernels.txt
import sqlite3, sys, glob
db = glob.glob("gpurun_out/prof_lt/**/lt_results.db", recursive=True)[0]
cur = sqlite3.connect(db).cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='view' or type='table'")]
kc = [d[1] for d in cur.execute("pragma table_info(kernels)")]
wg = next((c for c in kc if c.lower() in ("workgroup_size_x", "workgroup_size", "workgroup_x")), "0")
gr = next((c for c in kc if c.lower() in ("grid_size_x", "grid_size", "grid_x")), "0")
rows = cur.execute(f"select name, count(*), avg(end-start), max(vgpr_count), max(accum_vgpr_count), max(lds_size), max({wg}), max({gr}) from kernels group by name order by 3 desc").fetchall()
print("== NT", sys.argv[1])
for r in rows[:3]:
    print(f"  {r[2]/1e3:8.1f} us x{r[1]}  vgpr {r[3]} agpr {r[4]} lds {r[5]} wg {r[6]} grid {r[7]}  {r[0][:400]}")
PY
  rm -rf gpurun_out/prof_lt
done
cat gpurun_out/r04b_hipblaslt_kernels.txt
