#!/bin/bash
# workgroups-per-CU sweeps behind the grid sizes in the launchers: interleaved A/B inside one process per block
V='[{}, {"MI355_WG_PER_CU": "4"}, {"MI355_WG_PER_CU": "8"}, {"MI355_WG_PER_CU": "16"}, {"MI355_WG_PER_CU": "32"}, {"MI355_WG_PER_CU": "64"}]'
for case in filter65 fir65 filter3000 fft4096; do python tools/probe.py ab $case "$V"; done
