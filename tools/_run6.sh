export TMPDIR=/tmp
O=gpurun_out/r03_f; mkdir -p $O
(time timeout 2400 python -m pytest tests/test_gpu_prove.py::test_config5_prove_at_2_22_both_shapes tests/test_gpu_dist.py::test_config4_eight_ranks_share_the_gpu_at_2_20 tests/test_host_mirror.py::test_a_2_22_row_circuit_is_proved_and_accepted_by_the_pairing_verifier -x -q --durations=5) > $O/pytest.log 2>&1; tail -25 $O/pytest.log
