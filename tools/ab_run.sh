# usage (GPU box): bash tools/ab_run.sh "<command>"  -- the command once per variant library of tools/ab_variants.sh (the built library is put back afterwards)
ROOT=$(cd $(dirname $0)/.. && pwd); LIBD=$ROOT/vvcsoftware_vtm_amd/lib
cp $LIBD/libvvcgpu.so /tmp/libvvcgpu.keep.so
for f in $LIBD/variants/libvvcgpu.*.so; do
  NAME=$(basename $f .so | sed 's/libvvcgpu.//')
  cp $f $LIBD/libvvcgpu.so
  echo "== $NAME"; timeout -k 10 200 bash -c "$1" 2>&1 | grep -v amdgpu.ids
done
cp /tmp/libvvcgpu.keep.so $LIBD/libvvcgpu.so
