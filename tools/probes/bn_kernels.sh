# per-(kernel, grid) times of the fused BN kernels on the HRNet branch shapes
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
D=/tmp/bnk; rm -rf $D
rocprofv3 --kernel-trace --output-format csv -d $D -- python3 $ROOT/tools/bn_shapes.py > $D.log 2>&1
python3 $ROOT/tools/summarize_profile.py bygrid $(find $D -name '*kernel_trace.csv' | head -1) > /tmp/bnk.csv
python3 - <<'P'
import csv
for r in csv.reader(open('/tmp/bnk.csv')):
    if 'k_bn' in r[0]:
        print(r[0].replace('void ','').replace('(anonymous namespace)::','').split('(')[0], r[1], r[2], r[3], r[4], r[5], r[6])
P
