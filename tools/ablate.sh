#!/bin/bash
for d in 0 1 4 5; do echo "== ELD_CONV_DBG=$d"; ELD_CONV_DBG=$d python tools/layer_bench.py 2>&1 | grep -E "conv1_2|conv9_2|conv8_1|conv7_1|conv4_2|conv5_2"; done
