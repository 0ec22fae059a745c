#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06g; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
for M in stochastic host_trap; do
  if [ $M = stochastic ]; then U="--pc-sampling-unit cycles --pc-sampling-interval 65536"; else U="--pc-sampling-unit time --pc-sampling-interval 10"; fi
  MV_LIB_PATH=$R/megaverse_amd/_variants/libmv_lines.so timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $M $U --kernel-trace -d $OUT/pcs_$M -o pcs --output-format csv -- \
     python $R/bench.py --steps 160 --warmup 40 --no-cpu-baseline --profile-steps 0 --no-extra-legs > $OUT/pcs_$M.log 2>&1
  echo "== $M rc=$?"; tail -3 $OUT/pcs_$M.log | cut -c1-300
  find $OUT/pcs_$M -type f | head; 
  for f in $(find $OUT/pcs_$M -name "*pc_sampling*csv"); do wc -l $f; head -3 $f | cut -c1-400; done
done
# keep the merged output small: per-instruction histograms only
python - <<'PY'
import csv, glob, collections, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
for f in glob.glob(R+"/gpurun_out/r06g/pcs_*/**/*pc_sampling*csv", recursive=True):
    c=collections.Counter(); n=0
    with open(f) as fh:
        rd=csv.DictReader(fh)
        cols=rd.fieldnames
        for row in rd:
            n+=1
            c[(row.get("Instruction",""), row.get("Instruction_Comment",""), row.get("Dispatch_Id","") and "")]+=1
    out=f.replace(".csv","_hist.txt")
    with open(out,"w") as o:
        o.write("columns: %s\nsamples: %d\n"%(cols,n))
        for (ins,com,_),k in c.most_common(4000): o.write("%d\t%s\t%s\n"%(k,ins,com))
    os.remove(f)
    print(out, n)
PY
du -sh $OUT
