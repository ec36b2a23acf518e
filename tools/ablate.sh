#!/bin/bash
for d in 0 8; do echo "== ELD_CONV_DBG=$d"; ELD_CONV_DBG=$d python tools/layer_bench.py 2>&1 | grep -E "conv1_2|conv9_1|conv9_2|conv8_1|conv7_1|conv4_2|conv2_2|sum ms"; done
