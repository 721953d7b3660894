# SQ counters of the hidden-layer kernel by tile at 196 608 rows: tile 1 (8 waves, 176 x 64 per wave), tile 6 (4 waves x 512 registers, 128 x 128 per wave),
# tile 4 (4 waves x 512 registers, 176 x 128 per wave: spills in its K loop).  The program directly behind `--`.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d /tmp/prof_tiles -- python3 $R/tools/gemm_tile_ab.py --tiles 1,6,4 --rows 196608 --reps 3 > $O/r6_gemm_tiles_pmc.log 2>&1
python3 $R/tools/rocprof_summary.py pmc "$(find /tmp/prof_tiles -name '*counter_collection.csv' | head -1)" k_split_gemm > $O/r6_gemm_tiles_pmc.txt
grep "tile\": [146]" $O/r6_gemm_tiles_pmc.log >> $O/r6_gemm_tiles_pmc.txt
