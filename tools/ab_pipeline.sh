run() { python bench.py --legs train --no-fp32 --no-roofline --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step'],2))" >> gpurun_out/pipe_ab.log; }
python -m pytest tests/test_model_gpu.py -q -x -k "pipelined" 2>&1 | tail -5 > gpurun_out/pipe_ab.log
IM2IM_BWD_PIPELINE=0 run base
IM2IM_BWD_PIPELINE_MIN_BYTES=$((200<<20)) run min200M
IM2IM_BWD_PIPELINE_MIN_BYTES=$((400<<20)) run min400M
IM2IM_BWD_PIPELINE_MIN_BYTES=$((800<<20)) run min800M
IM2IM_BWD_PIPELINE_MIN_BYTES=0 run min0
IM2IM_WGRAD_STREAM=0 IM2IM_BWD_PIPELINE=0 run noside_nopipe
IM2IM_WGRAD_STREAM=0 IM2IM_BWD_PIPELINE=1 run noside_pipe
IM2IM_BWD_PIPELINE=0 run base
