for fr in 96000 441000 2646000; do for dt in i16 i32; do for e in A=1 HIPSOXR_NO_TILE_SPLIT=1; do
echo -n "44.1->16k frames $fr $dt $e: "; env $e DTYPE=$dt python tools/time_config.py 44100 16000 VHQ $fr 1 1 6 2>&1 | tail -1 | cut -c1-40
done; done; done
for fr in 96000 480000; do for dt in i32; do for e in A=1 HIPSOXR_NO_TILE_SPLIT=1; do
echo -n "48->44.1 frames $fr $dt $e: "; env $e DTYPE=$dt python tools/time_config.py 48000 44100 VHQ $fr 1 1 6 2>&1 | tail -1 | cut -c1-40
done; done; done
