# FETCH_SIZE / WRITE_SIZE of the bench line's roofline kernel alone (the head's 144 -> 720 weight gradient, wave-level splits)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  D=/tmp/hwp; rm -rf $D
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $D -- python3 $ROOT/tools/per_shape_roofline.py --only "fine part" > $D.log 2>&1
  python3 $ROOT/tools/summarize_profile.py pmc $(find $D -name '*counter_collection.csv' | head -1) | grep "k_wgrad3x3d\|k_wgrad_reduce\|k_conv3x3_il"
done
