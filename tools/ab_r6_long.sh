# Round 6, same-box A/B of library builds (make -C lm.rs_amd/csrc V=name EXTRA=-D...) on decode step time by position -> profiles/r6_ab_long_decode.txt.   usage: ab_r6_long.sh name...
cd $GRAFT_REPO_ROOT; O=gpurun_out/r6; mkdir -p $O
for n in base "$@"; do
  L=$PWD/lm.rs_amd/liblmrs_hip_$n.so; [ $n = base ] && L=$PWD/lm.rs_amd/liblmrs_hip.so
  echo "== $n"
  LMRS_LIB=$L timeout 300 python tools/decode_by_position.py 2>&1 | grep -E "positions +(392|520|776|1032|1288|1544|1800)"
done > $O/ab_long_decode.txt 2>&1
cat $O/ab_long_decode.txt
