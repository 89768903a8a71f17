export GPU_MAX_HW_QUEUES=8
timeout -k 10 2400 bash tools/capture_profiles.sh r06 2>&1 | tail -60
