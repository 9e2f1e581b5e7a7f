#!/bin/bash
# exact engine (kernel 6) product vs variant build, 48k -> 44.1k VHQ: sizes in 64-period slabs, float32 and int16
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
V=${1:-prevk}
for rep in 1 2; do
for slabs in ${SLABS:-10 47 282 768 6016}; do
  frames=$((slabs * 64 * 160))
  for dt in f32 i16; do
    echo -n "slabs64=$slabs $dt [product]: "; DTYPE=$dt python tools/time_config.py 48000 44100 VHQ $frames 1 1 6 2>&1 | tail -1 | cut -c1-30
    echo -n "slabs64=$slabs $dt [$V]: "; DTYPE=$dt tools/with_variant.sh $V python tools/time_config.py 48000 44100 VHQ $frames 1 1 6 2>&1 | tail -1 | cut -c1-30
  done
done; done
