#!/bin/bash
# which memory-side counters this box's rocprofv3 offers (MALL / DRAM / fabric): gpurun -- 'bash tools/list_counters.sh'
rocprofv3 --list-avail 2>/dev/null | grep -i -E "dram|hbm|mall|umc|EA0_RD|EA0_WR|TCC_REQ|TCC_MISS|TCC_HIT" | cut -c1-160 | head -60
