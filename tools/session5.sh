O=gpurun_out/s5; mkdir -p $O
timeout -k 10 60 ./build_tools/glds_test > $O/glds_test.txt 2>&1; cat $O/glds_test.txt
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "wide or big_tiles or conv_bn_stats_bf16 or adjoint" > $O/wide_tests.log 2>&1; tail -25 $O/wide_tests.log
