# usage (build container): bash tools/ab_variants.sh build <file.hip> NAME "-DDEFINE ..." [NAME2 "..."]...   -- variant libraries under vvcsoftware_vtm_amd/lib/variants/
#        (GPU box):        bash tools/ab_variants.sh run <stage> <kernel substring>                         -- tools/ktime.sh with each variant in place
# A/B timing aid: one source file compiled with different -D switches, every other object as built.
MODE=$1; shift
ROOT=$(cd $(dirname $0)/.. && pwd); LIBD=$ROOT/vvcsoftware_vtm_amd/lib
if [ "$MODE" = build ]; then
  SRC=$1; shift; BASE=$(basename $SRC .hip); mkdir -p $LIBD/variants
  while [ $# -gt 0 ]; do
    NAME=$1; DEFS=$2; shift 2
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-fast-math -ffp-contract=off -I$ROOT/include -I$ROOT/vvcsoftware_vtm_amd/csrc $DEFS -c $ROOT/vvcsoftware_vtm_amd/csrc/$SRC -o $LIBD/variants/$BASE.$NAME.o 2>/dev/null || { echo "compile failed: $NAME"; exit 1; }
    OBJS=$(ls $LIBD/obj/*.o | grep -v "/$BASE.o")
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIBD/variants/libvvcgpu.$NAME.so $OBJS $LIBD/variants/$BASE.$NAME.o && echo "built $NAME"
  done
else
  STAGE=$1; SUB=$2
  cp $LIBD/libvvcgpu.so /tmp/libvvcgpu.keep.so
  for f in $LIBD/variants/libvvcgpu.*.so; do
    NAME=$(basename $f .so | sed 's/libvvcgpu.//')
    cp $f $LIBD/libvvcgpu.so
    echo "== $NAME"; timeout -k 10 150 bash $ROOT/tools/ktime.sh $STAGE $SUB 2>&1 | tail -3
  done
  cp /tmp/libvvcgpu.keep.so $LIBD/libvvcgpu.so
fi
