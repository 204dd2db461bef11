set -o pipefail
timeout -k 10 1000 python -m pytest tests -m gpu -q > gpurun_out/fin_tests.log 2>&1; rc=$?; tail -4 gpurun_out/fin_tests.log; [ $rc -eq 0 ] || exit $rc
