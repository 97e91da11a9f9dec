# FETCH_SIZE of the BasicBlock weight gradients with / without the adjacent-strip grouping
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for G in 0 1; do
  export DCL_WGRAD_GROUP=$G
  D=/tmp/wgp; rm -rf $D
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D -- python3 $ROOT/tools/per_shape_roofline.py --only "b" > $D.log 2>&1
  echo "group=$G"
  python3 - <<'P'
import csv, glob, collections
f = glob.glob('/tmp/wgp/**/*counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'k_wgrad3x3d' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
        agg[(r['Kernel_Name'].split('(')[0][-30:], r['Grid_Size'])].append(float(r['Counter_Value']))
for k, v in sorted(agg.items()):
    print(k, len(v), round(sum(v) / len(v) / 1024, 1), 'MiB FETCH_SIZE raw')
P
done
