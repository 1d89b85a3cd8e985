python -m pytest tests/test_gpu_parity.py -x -q -k "heads or conv_rgb or narrow or train or conv3x3x3" 2>&1 | tail -4
python tools/train_launch_table.py > gpurun_out/r04_train_launch_table_b4.txt 2>&1; grep -E "scene|conv_wgrad +(2621440|2097152|1048576|655360)" gpurun_out/r04_train_launch_table_b4.txt
