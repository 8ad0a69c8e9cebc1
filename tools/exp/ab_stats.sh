R=$GRAFT_REPO_ROOT; O=/tmp/abstats; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export REPS=1 STEPS=1500
for rep in 1 2; do
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/r1_$rep -- python $R/_r1snap/rate.py > $O/r1_$rep.log 2>&1
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cur_$rep -- python $R/tools/exp/rate.py > $O/cur_$rep.log 2>&1
done
python - <<PY
import csv,glob
for rep in (1,2):
    fa=glob.glob('/tmp/abstats/r1_%d/*/*kernel_stats.csv'%rep)[0]; fb=glob.glob('/tmp/abstats/cur_%d/*/*kernel_stats.csv'%rep)[0]
    A={r[0]:float(r[3])/1e3 for r in list(csv.reader(open(fa)))[1:] if int(r[1])>1000}
    B={r[0]:float(r[3])/1e3 for r in list(csv.reader(open(fb)))[1:] if int(r[1])>1000}
    print("rep",rep); tot=0
    for k in A:
        if k in B: print("  %-90s r1 %.2f cur %.2f  d %+.2f"%(k[:90],A[k],B[k],B[k]-A[k])); tot+=B[k]-A[k]
    print("  sum of differences %.2f us; r1 sum %.2f cur sum %.2f"%(tot,sum(A.values()),sum(B.values())))
PY
