# Same-box A/B of two builds of the library: rdo-ptq_amd/lib/ab/librdoptq_hip_{old,new}.so (built beforehand, e.g. from a stash and
# from the working tree) are copied over lib/librdoptq_hip.so in turn and bench.py's sustained figure is printed, interleaved twice.
R=${GRAFT_REPO_ROOT:-.}
L=$R/rdo-ptq_amd/lib
mkdir -p $L/ab; cp $L/librdoptq_hip.so /tmp/lib_keep.so
for v in old new old new; do
  cp $L/ab/librdoptq_hip_$v.so $L/librdoptq_hip.so
  python3 $R/bench.py --no-cpu-baseline --no-extras --recon-iters 0 --sustain-steps 300 2>&1 >/dev/null | grep sustained | tail -1 | sed "s/^/$v /"
done
cp /tmp/lib_keep.so $L/librdoptq_hip.so
