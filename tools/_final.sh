export GPU_MAX_HW_QUEUES=8
timeout -k 5 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout -k 5 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout -k 10 1500 bash tools/capture_profiles.sh r06 2>&1 | tail -40
