cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
rm -rf $O/pmc_gemm
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_gemm -o g -- python3 $R/tools/gemm_time.py 4096 2900 > $O/pmc_gemm.log 2>&1
python3 - <<'PY'
import csv, collections, os
rows=list(csv.DictReader(open(os.environ.get('GRAFT_REPO_ROOT','/root/repo')+'/gpurun_out/pmc_gemm/g_counter_collection.csv')))
agg=collections.defaultdict(list)
for r in rows:
    if 'gemm' in r['Kernel_Name']:
        agg[(r['Kernel_Name'][:45], r['Grid_Size'])].append(float(r['Counter_Value']))
for k,v in agg.items():
    # group by distinct magnitude (M=4096 vs 2900)
    v=sorted(v); print(k, len(v), 'min %.2f GB max %.2f GB (raw FETCH_SIZE KB*1024)'%(v[0]*1024/1e9, v[-1]*1024/1e9))
PY
