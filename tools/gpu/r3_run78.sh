#!/bin/bash
mkdir -p gpurun_out
G12="tests/test_detection.py::test_g12_objectdetectionnet_hip_vs_reference"
timeout -k 10 300 python -m pytest $G12 -q -m gpu -s > gpurun_out/r78_a.log 2>&1; echo "alone rc=$?"; grep "G12 eval" gpurun_out/r78_a.log
timeout -k 10 400 python -m pytest tests/test_graph_gpu.py $G12 -q -m gpu -s > gpurun_out/r78_b.log 2>&1; echo "graph+G12 rc=$?"; grep "G12 eval" gpurun_out/r78_b.log
timeout -k 10 400 python -m pytest tests/test_vision_gpu.py $G12 -q -m gpu -s > gpurun_out/r78_c.log 2>&1; echo "vision+G12 rc=$?"; grep "G12 eval" gpurun_out/r78_c.log
timeout -k 10 400 python -m pytest tests/test_conv_gpu.py $G12 -q -m gpu -s > gpurun_out/r78_d.log 2>&1; echo "conv+G12 rc=$?"; grep "G12 eval" gpurun_out/r78_d.log
exit 0
