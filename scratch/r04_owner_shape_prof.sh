#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_ownerprof; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -o run -- python3 scratch/r04_owner_shape_time.py > $out/stdout.txt 2>$out/stderr.txt
python3 - <<'PY'
import csv,glob,re,collections
f=glob.glob("gpurun_out/r04_ownerprof/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
seq=[]
for r in rows:
    n=r["Kernel_Name"]
    m=re.search(r"(optimizer_ids_big_kernel|optimizer_kernel|lds_sort_kernel|merge_rank_kernel|part_sort_kernel)", n)
    if m: seq.append((m.group(1), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
# the script runs 6 (opt, fill) cases: 12 x (plan kernels + optimizer_kernel), then 12 x big kernel
cases=[("sgd",0.5),("sgd",0.25),("sgd",1.0),("adagrad",0.5),("adagrad",0.25),("adagrad",1.0)]
i=0; out=[]
import itertools
groups=[list(g) for k,g in itertools.groupby(seq, key=lambda x: x[0]=="optimizer_ids_big_kernel")]
gi=0
for c in cases:
    plan=groups[gi]; big=groups[gi+1]; gi+=2
    agg=collections.defaultdict(list)
    for k,v in plan: agg[k].append(v)
    per_step={k: sum(v)/12 for k,v in agg.items()}
    print(c, "plan+step per call:", {k: round(v,1) for k,v in per_step.items()}, "sum", round(sum(per_step.values()),1), "| one launch:", round(sorted(v for _,v in big)[len(big)//2],1))
PY
