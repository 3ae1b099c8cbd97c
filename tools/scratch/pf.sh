for n in 256 512; do
  for cfg in "0 128" "1 128" "1 64" "1 256"; do set -- $cfg
    echo "n=$n small_tiles=$1 min_wgs=$2: $(LMRS_GEMM_SMALL_TILES=$1 LMRS_GEMM_MIN_WGS=$2 python tools/prefill_rate.py llama-3.2-1b $n | head -1 | sed 's/.*MFMA): //')"
  done
done
