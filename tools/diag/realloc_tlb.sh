#!/bin/bash
# Address-translation counters of the surface kernel's launches while tools/diag/realloc_states.py frees and allocates the path
# state several times in one process (do the slow rounds miss the translation caches more?).  Run on the GPU box from the repository root.
root=$PWD
mkdir -p $root/gpurun_out/tlb
export TMPDIR=/tmp
cd /tmp
timeout -s KILL 280 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum --output-format csv -d $root/gpurun_out/tlb/out -- python3 $root/tools/diag/realloc_states.py 6 > $root/gpurun_out/tlb/run.log 2>&1
cd $root
tail -8 gpurun_out/tlb/run.log
python3 - <<'PY'
import csv, glob, collections
rows=[]
for f in glob.glob('gpurun_out/tlb/out/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_surface' in r['Kernel_Name']:
            rows.append((int(r['Dispatch_Id']), r['Counter_Name'], float(r['Counter_Value'])))
by=collections.defaultdict(dict)
for d,n,v in rows: by[d][n]=by[d].get(n,0)+v
big=[(d,v) for d,v in sorted(by.items()) if v.get('TCP_UTCL1_TRANSLATION_HIT_sum',0)>1e7]
for d,v in big[:60]:
    print(d, {k:int(x) for k,x in v.items()})
PY
rm -rf gpurun_out/tlb/out
