"""
Instruments k_mcts_select with 10-ns stamps after its phases: rewrites select_stats fields 4..7 to hold
[4] the expansion at the end of the fused kernel, [5] the top of the kernel (scalar loads, the children's backup by wave 0),
[6] + path staging and chains, [7] + float32 re-validation (pass A); [2] stays the whole of staging + re-validation, [3] the walk.
For measurements only --

    cp rl-rubiks_amd/csrc/rubiks_mcts.hip /tmp/keep.hip && python tools/select_stamps_patch.py && make -C rl-rubiks_amd
    python tools/tail_tree_phases.py f32s           # profiles/r3_tail_tree_phases.txt
    cp /tmp/keep.hip rl-rubiks_amd/csrc/rubiks_mcts.hip && make -C rl-rubiks_amd
"""
import os
p = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'rl-rubiks_amd', 'csrc', 'rubiks_mcts.hip')
s = open(p).read()


def rep(old, new):
    global s
    assert old in s, old
    s = s.replace(old, new, 1)


rep('''    const int nlev = plen_old - 1;     // levels 0 .. nlev - 1 carry an action; level nlev is the old leaf
''', '''    __shared__ int s_stamp[4];
    if (tid == 0) s_stamp[0] = (int)(wall_clock64() - t_begin);
    const int nlev = plen_old - 1;     // levels 0 .. nlev - 1 carry an action; level nlev is the old leaf
''')
rep('''    for (int k = tid; k < nlev; k += NT) s_next[k] = (u16)atomicExch(&s_head[sel_hash(s_node[k])], k);
    __syncthreads();
    if (!resume) {''', '''    for (int k = tid; k < nlev; k += NT) s_next[k] = (u16)atomicExch(&s_head[sel_hash(s_node[k])], k);
    __syncthreads();
    if (tid == 0) s_stamp[1] = (int)(wall_clock64() - t_begin);
    if (!resume) {''')
rep('''        __syncthreads();
        // Pass B: the flagged levels in float64''', '''        __syncthreads();
        if (tid == 0) s_stamp[2] = (int)(wall_clock64() - t_begin);
        // Pass B: the flagged levels in float64''')
rep('''            m.select_stats[8 * t + 4] = (int)(clock64() - c_walk);          // shader cycles of the walk''',
    '''            m.select_stats[8 * t + 4] = (int)(clock64() - c_walk);          // shader cycles of the walk
            s_stamp[3] = (int)(wall_clock64() - t_begin);''')
rep('''            m.select_stats[8 * t + 5] = slow_levels;
            m.select_stats[8 * t + 6] = revisits;
#endif
            m.select_stats[8 * t + 7] = (line_rounds << 16) | min(line_levels, 0xFFFF);''', '''            m.select_stats[8 * t + 5] = s_stamp[0];
            m.select_stats[8 * t + 6] = s_stamp[1];
#endif
            m.select_stats[8 * t + 7] = s_stamp[2];''')
rep('''        expand_leaf_wave(m, t, slot, lane, reinterpret_cast<const u8 *>(s_lut), max_states, false, cur);
}''', '''        expand_leaf_wave(m, t, slot, lane, reinterpret_cast<const u8 *>(s_lut), max_states, false, cur);
    if (lane == 0 && m.select_stats) m.select_stats[8 * t + 4] = (int)(wall_clock64() - t_begin) - s_stamp[3];   // expansion
}''')
open(p, 'w').write(s)
