# usage: bash tools/variant.sh <name> <file.hip> "<extra -D flags>"  -> fbpic_amd/csrc/libfbpic_amd_<name>.so
# (developer A/B builds: one object recompiled with extra flags, linked with the current others)
set -e
cd fbpic_amd/csrc
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -Wno-unused-result -Wno-pass-failed -I../../include"
/opt/rocm/bin/hipcc $F $3 -c $2 -o /tmp/var_$1.o 2>/dev/null
OBJS=""
for o in runtime comm particles sort handover deposit fields fft zfft hankel; do
  if [ "$o.hip" = "$2" ]; then OBJS="$OBJS /tmp/var_$1.o"; else OBJS="$OBJS $o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libfbpic_amd_$1.so $OBJS -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
