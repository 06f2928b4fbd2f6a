cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s26; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pv -- python3 $GRAFT_REPO_ROOT/tools/gemm_vendor_compare.py > $O/vendor.txt 2>&1)
cp $(find /tmp/pv -name "*kernel_stats.csv" | head -1) $O/vendor_kernel_stats.csv
python3 - <<'PY'
import csv,glob,os
f=glob.glob("/tmp/pv/**/*kernel_trace.csv",recursive=True)[0]
seen={}
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if n not in seen:
        seen[n]={k:r[k] for k in r if k in ("LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Workgroup_Size_X","Grid_Size_X","Grid_Size_Y","Grid_Size_Z","Workgroup_Size_Y")}
        seen[n]["n"]=0
    seen[n]["n"]+=1
out=open(os.environ["GRAFT_REPO_ROOT"]+"/gpurun_out/s26/vendor_kernels.txt","w")
for n,v in seen.items():
    out.write(n[:300]+"\n   "+str(v)+"\n")
PY
cat $O/vendor_kernels.txt | head -80
