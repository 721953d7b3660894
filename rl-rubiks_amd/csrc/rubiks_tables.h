// Compile-time construction of the 3x3x3 move tables in the layouts the gfx950 kernels stage
// into LDS.  Restates what the reference builds at import time
// (librubiks/cube/maps.py:74-98 `Actions`, :107-145 `get_tensor_map`; used at librubiks/cube/cube.py:239).
#pragma once
#include <stdint.h>

namespace rubiks {

constexpr int kPlanes = 20;    // 8 corner cubies then 12 edge cubies
constexpr int kCorners = 8;
constexpr int kActions = 12;   // a = 2*face + (1 - direction); faces F,B,T,D,L,R
constexpr int kCodes = 24;     // corner: 3*pos+ori, edge: 2*pos+ori
constexpr int kCodePad = 32;   // tables are padded to 5-bit codes so that any byte & 31 stays in range
constexpr int kActionPad = 16; // ... and any action & 15

struct FaceMove {
    int corner_cycle[4];   // positions, positive revolution: cycle[i] -> cycle[i+1]
    int edge_cycle[4];
    int corner_keep;       // the corner orientation that is preserved; the other two trade places
    bool edge_flip;        // edge orientation toggles
};

// maps.py:74-98
constexpr FaceMove kFaceMoves[6] = {
    /*F*/ {{0, 1, 2, 3}, {0, 1, 2, 3}, 0, false},
    /*B*/ {{4, 7, 6, 5}, {8, 11, 10, 9}, 0, false},
    /*T*/ {{0, 3, 7, 4}, {0, 7, 8, 4}, 1, true},
    /*D*/ {{1, 5, 6, 2}, {2, 5, 10, 6}, 1, true},
    /*L*/ {{0, 4, 5, 1}, {1, 4, 9, 5}, 2, false},
    /*R*/ {{7, 3, 2, 6}, {3, 6, 11, 7}, 2, false},
};

struct MoveTables {
    // lut[a][kind][v]: code v after action a.  Entries with a >= 12 or v >= 24 are identity padding.
    uint8_t lut[kActionPad][2][kCodePad];
    // lut4[kind][v][a]: the same numbers action-minor, so that one 4-byte LDS read returns the codes
    // of four consecutive children (used by the 12-child expansion).
    uint8_t lut4[2][kCodePad][kActions];
    int8_t solved[kPlanes];
};

constexpr MoveTables make_tables() {
    MoveTables t{};
    for (int a = 0; a < kActionPad; ++a)
        for (int k = 0; k < 2; ++k)
            for (int v = 0; v < kCodePad; ++v) t.lut[a][k][v] = (uint8_t)v;
    for (int f = 0; f < 6; ++f) {
        const FaceMove &m = kFaceMoves[f];
        const int a_pos = 2 * f, a_neg = 2 * f + 1;   // direction 1 (positive) is the even action
        for (int j = 0; j < 4; ++j) {
            const int cf = m.corner_cycle[j], ct = m.corner_cycle[(j + 1) & 3];
            for (int o = 0; o < 3; ++o) {
                const int no = (o == m.corner_keep) ? o : 3 - m.corner_keep - o;   // maps.py:128
                const int from = 3 * cf + o, to = 3 * ct + no;
                t.lut[a_pos][0][from] = (uint8_t)to;
                t.lut[a_neg][0][to] = (uint8_t)from;   // negative turn = inverse (maps.py:132)
            }
            const int ef = m.edge_cycle[j], et = m.edge_cycle[(j + 1) & 3];
            for (int o = 0; o < 2; ++o) {
                const int no = m.edge_flip ? (o ^ 1) : o;   // maps.py:135
                const int from = 2 * ef + o, to = 2 * et + no;
                t.lut[a_pos][1][from] = (uint8_t)to;
                t.lut[a_neg][1][to] = (uint8_t)from;
            }
        }
    }
    for (int k = 0; k < 2; ++k)
        for (int v = 0; v < kCodePad; ++v)
            for (int a = 0; a < kActions; ++a) t.lut4[k][v][a] = t.lut[a][k][v];
    for (int i = 0; i < kCorners; ++i) t.solved[i] = (int8_t)(3 * i);          // cube.py:58-65
    for (int i = 0; i < 12; ++i) t.solved[kCorners + i] = (int8_t)(2 * i);
    return t;
}

constexpr MoveTables kTables = make_tables();

}  // namespace rubiks
