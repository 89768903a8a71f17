timeout -k 5 120 python tools/local_timeline_vol.py 12 blobs
timeout -k 5 120 python tools/local_timeline_vol.py 12 random
timeout -k 5 120 python tools/local_timeline_vol.py 4 blobs
