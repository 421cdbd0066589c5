"""The order the dense top of the cyclic reduction is stored in for the round's vector-ALU mat-vec (pympc_amd/csrc/mpcqp_topv.h): factor_bcr writes the inverse
entry by entry through bcr_topv_pos, the round's LDS copy is read pair by pair as bcr_topv_rc enumerates them -- the two maps must be inverse to each other
for every top size the schedules produce (2, 5, 7 stages) and both lane layouts.  Plain host functions: compiled here with g++ (no GPU)."""
import os
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = r'''
#include <cstdio>
#include <initializer_list>
#include "mpcqp_topv.h"
int main() {
    int bad = 0;
    for (int nt : {2, 5, 7}) {
        const int pairs = nt * nt * 128;                 // 16 nt rows x 16 nt columns, two doubles per pair
        for (int p = 0; p < pairs; ++p) {
            int row, col; bcr_topv_rc(nt, p, row, col);
            if (row < 0 || row >= 16 * nt || col < 0 || col + 1 >= 16 * nt + 1 || (col & 1)) ++bad;
            if (bcr_topv_pos(nt, row, col) != 2LL * p || bcr_topv_pos(nt, row, col + 1) != 2LL * p + 1) ++bad;
        }
        for (int row = 0; row < 16 * nt; ++row) for (int col = 0; col < 16 * nt; ++col) {      // onto: every entry has a place below 2 * pairs
            const long long q = bcr_topv_pos(nt, row, col);
            if (q < 0 || q >= 2LL * pairs) ++bad;
        }
    }
    std::printf("%d\n", bad);
    return bad != 0;
}
'''


@pytest.mark.parametrize('mode', [1, 2])
def test_position_map_inverts_the_pair_enumeration(mode):
    with tempfile.TemporaryDirectory() as d:
        src, exe = os.path.join(d, 't.cpp'), os.path.join(d, 't')
        open(src, 'w').write(SRC)
        subprocess.check_call(['g++', '-std=c++17', '-O1', '-DLATW_TOP_VALU=%d' % mode, '-I', os.path.join(ROOT, 'pympc_amd', 'csrc'), src, '-o', exe])
        out = subprocess.run([exe], capture_output=True, text=True)
        assert out.returncode == 0 and out.stdout.strip() == '0', out.stdout
