out=gpurun_out/${1:-r05n}; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_tokens.py tests/test_documents.py -x -q -m gpu > $out/gpu_tests_tokens.txt 2>&1 || { tail -40 $out/gpu_tests_tokens.txt; exit 1; }
tail -2 $out/gpu_tests_tokens.txt
timeout -k 10 150 python tests/stress_tokens.py 60 304 > $out/stress_tokens_304.txt 2>&1; tail -1 $out/stress_tokens_304.txt
timeout -k 10 120 python tests/stress_documents.py 30 305 > $out/stress_documents_305.txt 2>&1; tail -1 $out/stress_documents_305.txt
for w in minified utf8 pretty4; do
  python scripts/prep_prof.py $w 2>&1 | grep -v amdgpu | tail -1 | tee -a $out/rates.txt
  python scripts/prep_prof.py $w --match 2>&1 | grep -v amdgpu | tail -1 | tee -a $out/rates.txt
  python scripts/prep_prof.py $w --pairs 2>&1 | grep -v amdgpu | tail -1 | tee -a $out/rates.txt
done
[ -n "$2" ] && MATCH=1 bash scripts/prep_ab.sh libs minified $2 2>&1 | grep -v amdgpu | tee $out/ab_match.txt
(bash scripts/split_any.sh "minified --match"; bash scripts/split_any.sh "minified --pairs") 2>&1 | grep -E "^==|msj_tokens" | tee $out/split.txt
