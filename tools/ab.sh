# usage: bash tools/ab.sh "<kbench args>" [variant names...]  -> kbench with the current build and each
# fbpic_amd/csrc/libfbpic_amd_<variant>.so
ARGS="$1"
echo "== current"; python tools/kbench.py $ARGS 2>/dev/null
shift
for v in "$@"; do
  echo "== $v"; FBPIC_AMD_LIB=$PWD/fbpic_amd/csrc/libfbpic_amd_$v.so python tools/kbench.py $ARGS 2>/dev/null
done
