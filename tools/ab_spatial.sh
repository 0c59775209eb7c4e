for v in compact compact2; do
  HIKARI_HIP_LIB=$PWD/build_ab/$v.so timeout 400 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "cornell or named or spatial" 2>&1 | tail -2
done
bash tools/ab_variants.sh "2" base compact compact2 base compact compact2
bash tools/ab_variants.sh "5 4" base compact compact2
