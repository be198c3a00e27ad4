# usage: bash tools/variant.sh <name> <file.hip> "<extra flags>"  -> fbpic_amd/csrc/variants/libfbpic_amd_<name>.so
# (developer A/B builds: one object recompiled with extra flags, linked with the current others)
set -e
cd /root/repo/fbpic_amd/csrc
mkdir -p variants
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed -I../../include"
EXTRA=""
[ "$2" = "cycle.hip" ] && EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-sched-strategy=max-memory-clause"
/opt/rocm/bin/hipcc $F $EXTRA $3 -c $2 -o /tmp/var_$1.o
OBJS=""
for s in $(grep '^SRCS' Makefile | cut -d= -f2); do
  o=${s%.hip}
  if [ "$s" = "$2" ]; then OBJS="$OBJS /tmp/var_$1.o"; else OBJS="$OBJS $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/libfbpic_amd_$1.so $OBJS -L/opt/rocm/lib -lrocfft -ldl -Wl,-rpath,/opt/rocm/lib
