python -m pytest tests -m gpu -q -x 2>&1 | tail -12 > gpurun_out/r3_call6_pytest.txt
tail -4 gpurun_out/r3_call6_pytest.txt
if grep -q " failed" gpurun_out/r3_call6_pytest.txt; then echo "TESTS FAILED: profiles skipped"; exit 0; fi
bash tools/collect_profiles_r03.sh
