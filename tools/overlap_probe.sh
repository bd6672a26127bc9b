# diagnostics: duration of the cfg-5 kernels against the batch size (first full passes of a solve)
cd /tmp && export TMPDIR=/tmp
for B in 128 256 512 1024; do
O=$GRAFT_REPO_ROOT/gpurun_out/ovl$B; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/profile_cfg5.py $B 2 > $O/log 2>&1
tail -1 $O/log
python - $O <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for name in ('chain2', 'stage', 'sweep_kernel'):
    d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows if name in r['Kernel_Name'])
    print('   ', name, 'max', d[-1], 'p75', d[int(len(d) * .75)], 'median', d[len(d) // 2])
PY
done
