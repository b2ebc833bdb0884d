#!/usr/bin/env bash
# visit 21: where does the refill cost of the 1x1 convs come from?  ablations + input row strides around 2 KB
mkdir -p gpurun_out; G=gpurun_out
timeout 300 python tools/conv_probe.py 32:512:512:1:1:0 32:512:512:1:1:7 32:512:512:1:1:8 32:512:512:1:1:9 \
  32:480:512:1:1 32:496:512:1:1 32:504:512:1:1 32:508:512:1:1 32:512:512:1:1 32:516:512:1:1 32:520:512:1:1 32:528:512:1:1 32:544:512:1:1 32:576:512:1:1 \
  32:256:256:3:1:0 32:256:256:3:1:7 32:256:256:3:1:8 32:256:256:3:1:9 32:248:256:3:1 32:264:256:3:1 \
  32:512:512:1:0 32:520:512:1:0 32:512:256:1:1 32:520:256:1:1 32:256:512:1:1 32:264:512:1:1 > $G/v21_probe.md 2>&1
DR_CONV_GLDS=0 timeout 300 python tools/conv_probe.py 32:512:512:1:1 32:520:512:1:1 32:256:256:3:1 32:264:256:3:1 > $G/v21_probe_noglds.md 2>&1
cat $G/v21_probe.md $G/v21_probe_noglds.md
