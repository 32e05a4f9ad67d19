"""Host-side models of two index schemes the HIP kernels rely on (no GPU, no kernel call): they pin
the arithmetic the kernels' comments argue with, so that a change of a constant in csrc/ has a
test to answer to.

* the LDS plane layout of ``gemm_tn_split_kernel`` (csrc/gemm.hip): 20 dwords per staged column,
  lane -> (column quad, row group) map of the staging threads, fragment addresses of the readers;
  conflict-freedom is checked against the lane groups and bank moduli of MI355X_MICROARCH.md
  (ds_write_b128: 8 x 8 contiguous lanes, 32 banks; ds_read_b128: 4 x 16 lanes, 64 banks);
* the radix sort of csrc/graph.hip (per-tile digit counts, exclusive scan of the
  [digit][tile] table, stable placing pass) re-stated with numpy on small tiles: equal to a stable
  argsort for every pass count.
"""
import numpy as np
import pytest

K_CLD = 20          # dwords per staged column (kCLD)
K_TILE = 128        # columns per operand tile (kWTile)

# ds_read_b128 lane groups (MI355X_MICROARCH.md, LDS table)
READ_GROUPS = [
    [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27],
    [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31],
    [32, 33, 34, 35, 44, 45, 46, 47, 52, 53, 54, 55, 56, 57, 58, 59],
    [36, 37, 38, 39, 40, 41, 42, 43, 48, 49, 50, 51, 60, 61, 62, 63],
]


def _staging(lane, wave):
    """(first tile column, row group) of a staging thread — `tc`, `rg` in the kernel."""
    c4 = (lane & 1) + 2 * (lane >> 3)
    rg = (lane >> 1) & 3
    return 64 * (wave & 1) + 4 * c4, rg


def test_wgrad_planes_every_slot_is_written_once_and_read_where_it_was_written():
    # one (operand, term) plane: 128 columns x 16 dwords (32 rows as bf16 pairs) + padding
    owner = {}
    for wave in (0, 1):                       # the two waves that stage one operand
        for lane in range(64):
            tc, rg = _staging(lane, wave)
            for cc in range(4):
                base = (tc + cc) * K_CLD + 4 * rg      # 16-byte vector: rows 8 rg .. 8 rg + 7
                for dw in range(4):
                    key = base + dw
                    assert key not in owner, 'two threads write the same dword'
                    owner[key] = (tc + cc, 8 * rg + 2 * dw)   # column, first row of the pair
    assert len(owner) == K_TILE * 16
    # reader: lane (li, lh) of block i, step s reads 16 bytes at (col)*20 + 4 lh + 8 s and expects
    # rows 16 s + 8 lh .. + 7 of column `col`
    for i in range(2):
        for s in range(2):
            for lane in range(64):
                li, lh = lane & 31, lane >> 5
                col = i * 32 + li
                addr = col * K_CLD + 4 * lh + 8 * s
                for dw in range(4):
                    assert owner[addr + dw] == (col, 16 * s + 8 * lh + 2 * dw)


def test_wgrad_planes_are_bank_conflict_free_for_stores_and_loads():
    # stores: groups of 8 contiguous lanes, bank = dword address mod 32, 4 banks per lane
    for wave in (0, 1):
        for cc in range(4):
            for g in range(8):
                banks = []
                for lane in range(8 * g, 8 * g + 8):
                    tc, rg = _staging(lane, wave)
                    a = (tc + cc) * K_CLD + 4 * rg
                    banks += [(a + d) % 32 for d in range(4)]
                assert sorted(banks) == list(range(32)), (wave, cc, g)
    # loads: the four 16-lane groups, bank = dword address mod 64
    for s in range(2):
        for i in range(2):
            for wcol in (0, 64):
                for group in READ_GROUPS:
                    banks = []
                    for lane in group:
                        li, lh = lane & 31, lane >> 5
                        a = (wcol + i * 32 + li) * K_CLD + 4 * lh + 8 * s
                        banks += [(a + d) % 64 for d in range(4)]
                    assert sorted(banks) == list(range(64)), (s, i, wcol, group[0])


def test_wgrad_staging_loads_cover_full_lines():
    """A load instruction of a staging wave (fixed row-in-group r) fetches 4 rows x 256 contiguous
    bytes: every 128-byte line it touches is used completely."""
    lines = {}
    for lane in range(64):
        tc, rg = _staging(lane, 0)
        for b in range(16):                       # the lane's 16 bytes
            byte = tc * 4 + b
            lines.setdefault((rg, byte // 128), set()).add(byte % 128)
    assert len(lines) == 8 and all(len(v) == 128 for v in lines.values())


def _radix_sort_model(keys, bits, tile):
    """csrc/graph.hip, pass by pass: counts[digit][tile], exclusive scan in digit-major order,
    stable placement inside the tile in position order."""
    keys = np.asarray(keys, dtype=np.uint64)
    n = keys.size
    vals = np.arange(n, dtype=np.int64)
    if n == 0:  # (pygamd_index_sort returns before any launch)
        return keys, vals
    tiles = (n + tile - 1) // tile
    for p in range((bits + 7) // 8):
        digit = ((keys >> np.uint64(8 * p)) & np.uint64(255)).astype(np.int64)
        counts = np.zeros((256, tiles), dtype=np.int64)
        for t in range(tiles):
            d = digit[t * tile:(t + 1) * tile]
            counts[:, t] = np.bincount(d, minlength=256)
        starts = np.concatenate([[0], np.cumsum(counts.reshape(-1))[:-1]]).reshape(256, tiles)
        out_k, out_v = np.empty_like(keys), np.empty_like(vals)
        nxt = starts.copy()
        for t in range(tiles):
            for i in range(t * tile, min((t + 1) * tile, n)):   # position order = stable
                d = digit[i]
                out_k[nxt[d, t]] = keys[i]
                out_v[nxt[d, t]] = vals[i]
                nxt[d, t] += 1
        keys, vals = out_k, out_v
    return keys, vals


@pytest.mark.parametrize('n,hi,tile', [(0, 10, 16), (1, 1, 16), (37, 5, 16), (1000, 300, 64),
                                       (1000, 70000, 64), (513, 2 ** 20, 32)])
def test_radix_sort_model_is_a_stable_sort(n, hi, tile):
    rng = np.random.default_rng(n + hi)
    keys = (rng.random(n) ** 3 * hi).astype(np.uint64)
    bits = max(int(hi).bit_length(), 1)
    k, v = _radix_sort_model(keys, bits, tile)
    order = np.argsort(keys, kind='stable')
    assert np.array_equal(k, keys[order]) and np.array_equal(v, order)
