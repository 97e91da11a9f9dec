cd tools
for cc in 0 4 8 12 16 24; do echo "chunk $cc"; DCL_UPCE_BWD_CHUNK=$cc python upce_time.py 2>&1 | tail -3; done
