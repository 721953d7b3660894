"""
Instruments k_mcts_select with 10-ns stamps after its phases (children's backup + staging | float32 re-validation | float64 levels):
rewrites select_stats fields 5..7 to hold them instead of the revisit / line-following counters.  For measurements only --

    cp rl-rubiks_amd/csrc/rubiks_mcts.hip /tmp/keep.hip && python tools/select_stamps_patch.py && make -C rl-rubiks_amd
    python tools/select_pool_stats.py bf16          # profiles/r2f_select_pool_phases.txt
    cp /tmp/keep.hip rl-rubiks_amd/csrc/rubiks_mcts.hip && make -C rl-rubiks_amd
"""
import os
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rl-rubiks_amd', 'csrc', 'rubiks_mcts.hip')
s=open(p).read()
def rep(old,new):
    global s
    assert old in s, old
    s=s.replace(old,new,1)
rep('''    if (!resume) {
        // Pass A: one lane per level''','''    __shared__ int s_stamp[4];
    if (tid == 0) s_stamp[0] = (int)(wall_clock64() - t_begin);
    if (!resume) {
        // Pass A: one lane per level''')
rep('''        __syncthreads();
        // Pass B: the flagged levels in float64''','''        __syncthreads();
        if (tid == 0) s_stamp[1] = (int)(wall_clock64() - t_begin);
        // Pass B: the flagged levels in float64''')
rep('''        const int first = s_first;
''','''        if (tid == 0) s_stamp[2] = (int)(wall_clock64() - t_begin);
        const int first = s_first;
''')
rep('''            m.select_stats[8 * t + 5] = slow_levels;
            m.select_stats[8 * t + 6] = revisits;
            m.select_stats[8 * t + 7] = (line_rounds << 16) | min(line_levels, 0xFFFF);''','''            m.select_stats[8 * t + 5] = s_stamp[0];
            m.select_stats[8 * t + 6] = s_stamp[1];
            m.select_stats[8 * t + 7] = s_stamp[2];''')
open(p,'w').write(s)
