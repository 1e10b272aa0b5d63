# round 6, encode A/B blocks I / J (profiles/r06_encode_ab.txt): n2 := ||vh||^2 (no exponent mask) and the exact rescoring as packed FMAs
mkdir -p gpurun_out/r06
python tools/ab_time.py product tools/exp/libgq_n2norm.so tools/exp/libgq_pkfma.so tools/exp/libgq_pkfma_n2norm.so > gpurun_out/r06/encode_ab_IJ.txt 2>&1
for l in n2norm pkfma pkfma_n2norm; do
  echo "== $l: fixups and fuzz" >> gpurun_out/r06/encode_ab_IJ.txt
  GQ_LIB_PATH=$PWD/tools/exp/libgq_$l.so python tools/exp_time.py >> gpurun_out/r06/encode_ab_IJ.txt 2>&1
  GQ_LIB_PATH=$PWD/tools/exp/libgq_$l.so python tools/fuzz_prefilter.py 1500 77 2>&1 | tail -2 >> gpurun_out/r06/encode_ab_IJ.txt
done
python tools/exp_time.py >> gpurun_out/r06/encode_ab_IJ.txt 2>&1
GQ_AB_D=32 python tools/ab_time.py product tools/exp/libgq_n2norm.so tools/exp/libgq_pkfma_n2norm.so >> gpurun_out/r06/encode_ab_IJ.txt 2>&1
cat gpurun_out/r06/encode_ab_IJ.txt
