"""CPU models of two LDS layouts of csrc/gemm.hip whose address arithmetic cannot be exercised without a GPU: they restate the kernel's
formulas (cited below) on integer arrays and check what the kernel relies on -- every MFMA fragment gathers exactly the operand
elements it stands for, and every wave-level read touches each LDS bank once.

  * row-major stages (`tile_off`, common.h): `[rows][64 x 16 bit]` = 128-byte rows, 16-byte chunk c of row r stored at slot
    c ^ ((r >> 1) & 7); fragments by `ds_read_b128` (16x16x32: lane (row = lane & 15, kgroup = lane >> 4) reads chunk kgroup of a k-half).
  * K-major stages (gemm.hip "K-MAJOR operands", TAG == 2): `[64 k][BM]` rows of BM x 2 bytes, 32-byte segment g of stage row k stored at
    slot g ^ gsw(k), gsw(k) = (k & 3) | ((k >> 3) & 1) << 2; fragments by two `ds_read_b64_tr_b16` per 8-k operand
    (tools/probes/tr_read_probe.py pinned the instruction: a 16-lane group reads a [4][16] block, lane i supplies row i / 4, columns
    4 (i % 4) .. + 3, and receives column i)."""
import numpy as np
import pytest


def swz(r):                      # common.h swz(): the XOR term of a 128-byte row
    return (r >> 1) & 7


def test_row_major_stage_fragments_and_banks():
    rows = 128
    # logical tile T[r][k], k = 0..63, one 16-bit element = its own id
    T = np.arange(rows * 64, dtype=np.int64).reshape(rows, 64)
    lds = np.full(rows * 64, -1, dtype=np.int64)                       # element index inside the stage
    # staging (glds_slab_off callers): lane (lane >> 3 = row in the 8-row slab, lane & 7 = LDS chunk slot) copies SOURCE chunk slot ^ swz(r)
    for r in range(rows):
        for slot in range(8):
            c = slot ^ swz(r)
            lds[r * 64 + slot * 8: r * 64 + slot * 8 + 8] = T[r, c * 8: c * 8 + 8]
    assert (lds >= 0).all()
    # fragment read of one 16-row sub-tile, k-half kh: lane reads 16 bytes at tile_off(row, kh * 4 + kgroup)
    for sub in range(rows // 16):
        for kh in range(2):
            banks = []
            for lane in range(64):
                row, kg = sub * 16 + (lane & 15), lane >> 4
                chunk = kh * 4 + kg
                off = row * 128 + ((chunk ^ swz(row)) << 4)            # tile_off(), bytes
                got = lds[off // 2: off // 2 + 8]
                assert np.array_equal(got, T[row, chunk * 8: chunk * 8 + 8])      # the lane's 8 k-values of its row
                banks.append(((off // 4) % 64, lane))
            # ds_read_b128 is serviced 16 lanes at a time (MI355X_MICROARCH "LDS": 4 x 16-lane groups), each lane 4 banks wide
            for grp in ([0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]):
                for half in (0, 32):
                    used = set()
                    for lane in grp:
                        b0 = banks[lane + half][0]
                        for q in range(4):
                            assert (b0 + q) % 64 not in used
                            used.add((b0 + q) % 64)
                    assert len(used) == 64


def gsw(k):                      # gemm.hip: segment swizzle of a K-major stage row
    return (k & 3) | (((k >> 3) & 1) << 2)


@pytest.mark.parametrize("BM", [128, 256])
def test_k_major_stage_fragments_and_banks(BM):
    pitch = BM * 2                                                     # bytes per stage row (one k)
    T = np.arange(64 * BM, dtype=np.int64).reshape(64, BM)             # logical operand T[k][m]
    lds = np.full(64 * BM, -1, dtype=np.int64)
    # staging (setup_a / setup_w): a wave instruction covers RPI stage rows of LPR 16-byte chunk slots; slot s receives SOURCE chunk
    # s ^ (gsw(k) << 1)
    LPR = BM // 8
    for k in range(64):
        for slot in range(LPR):
            c = slot ^ (gsw(k) << 1)
            assert 0 <= c < LPR
            lds[k * BM + slot * 8: k * BM + slot * 8 + 8] = T[k, c * 8: c * 8 + 8]
    assert (lds >= 0).all()
    for seg in range(BM // 16):                                        # 16-column sub-tile (wm * WM / 16 + j in the kernel)
        for kh in range(2):
            frag = np.zeros((64, 8), dtype=np.int64)
            for half in range(2):                                      # the two transpose reads of an 8-k operand (rows + 0 and + 4)
                addr = []
                for lane in range(64):
                    ti, kg = lane & 15, lane >> 4
                    gl = (ti >> 2) | ((kg & 1) << 2)
                    off = (kg * 8 + (ti >> 2)) * pitch + ((seg ^ gl) * 32) + (ti & 3) * 8      # a_tr[j] / w_tr[i]
                    off += (kh * 32 + half * 4) * pitch                                        # ldfrag_t()
                    addr.append(off)
                # the instruction: per 16-lane group a [4 rows][16 cols] block, lane i <- column i of the 4 rows the group's lanes address
                for grp in range(4):
                    blk = np.zeros((4, 16), dtype=np.int64)
                    for i in range(16):
                        a = addr[grp * 16 + i]
                        blk[i >> 2, 4 * (i & 3): 4 * (i & 3) + 4] = lds[a // 2: a // 2 + 4]
                    for i in range(16):
                        frag[grp * 16 + i, half * 4: half * 4 + 4] = blk[:, i]
                # banks: 8 bytes per lane, 32 lanes per LDS pass
                for h32 in (0, 32):
                    used = set()
                    for lane in range(h32, h32 + 32):
                        for q in range(2):
                            b = (addr[lane] // 4 + q) % 64
                            assert b not in used, (BM, seg, kh, half, lane)
                            used.add(b)
                    assert len(used) == 64
            # MFMA 16x16x32 operand: lane (m = lane & 15, kg = lane >> 4) holds k = kh * 32 + kg * 8 .. + 7 of column seg * 16 + m
            for lane in range(64):
                m, kg = seg * 16 + (lane & 15), lane >> 4
                want = T[kh * 32 + kg * 8: kh * 32 + kg * 8 + 8, m]
                assert np.array_equal(frag[lane], want), (BM, seg, kh, lane)


# ---- work-item maps (index arithmetic only) --------------------------------------------------------------------------------------
def _tile_coords(t, tiles_m, tiles_n, group_m=8):       # gemm.hip tile_coords(): grouped rasterisation of the persistent schedule
    per_group = group_m * tiles_n
    gid = t // per_group
    first_m = gid * group_m
    gsz = min(tiles_m - first_m, group_m)
    r = t - gid * per_group
    return first_m + r % gsz, r // gsz


@pytest.mark.parametrize("tiles_m,tiles_n,group_m", [(127, 16, 8), (57, 8, 8), (1, 9, 8), (29, 4, 8), (15, 4, 3), (8, 24, 8), (113, 1, 8)])
def test_gemm_tile_rasterisation_is_a_bijection(tiles_m, tiles_n, group_m):
    seen = set()
    for t in range(tiles_m * tiles_n):
        mb, nb = _tile_coords(t, tiles_m, tiles_n, group_m)
        assert 0 <= mb < tiles_m and 0 <= nb < tiles_n
        seen.add((mb, nb))
    assert len(seen) == tiles_m * tiles_n
    # consecutive work items of a group share the W panel (same nb) group_m at a time: what keeps the panel in one XCD's L2
    mb0, nb0 = _tile_coords(0, tiles_m, tiles_n, group_m)
    mb1, nb1 = _tile_coords(1, tiles_m, tiles_n, group_m) if tiles_m * tiles_n > 1 else (mb0, nb0)
    assert nb1 == nb0 or min(tiles_m, group_m) == 1


def _xcd_head_map(L, nb, BH):                           # attention.hip / attention_bwd2.hip: (block of a head, head) of linear workgroup L
    blk, bh = L % nb, L // nb
    if BH % 8 == 0:
        xcd, idx = L & 7, L >> 3
        blk, bh = idx % nb, (idx // nb) * 8 + xcd
    return blk, bh


@pytest.mark.parametrize("nb,BH", [(8, 576), (15, 16), (8, 16), (4, 128), (29, 8), (8, 12), (1, 24)])
def test_attention_block_map_keeps_a_head_on_one_xcd(nb, BH):
    seen = {}
    for L in range(nb * BH):
        blk, bh = _xcd_head_map(L, nb, BH)
        assert 0 <= blk < nb and 0 <= bh < BH
        assert (blk, bh) not in seen
        seen[(blk, bh)] = L
    assert len(seen) == nb * BH
    if BH % 8 == 0:      # workgroups are dealt to the 8 XCDs round-robin by linear id: all blocks of a head must share L & 7
        for bh in range(BH):
            assert len({seen[(blk, bh)] & 7 for blk in range(nb)}) == 1


# ---- upsample4x_planes_scaled_kernel (elementwise.hip): bands, output-row shares and the rolling source rows --------------------------
def _src_tap(r, i, n):
    """common.h src_tap in float32: i0 = floor(r i), i1 = min(i0 + 1, n - 1)"""
    s = np.float32(r) * np.float32(i)
    i0 = int(s)
    return i0, i0 + (1 if i0 < n - 1 else 0)


@pytest.mark.parametrize("H,W,LB", [(120, 120, 16), (120, 120, 8), (32, 24, 16), (15, 20, 16), (6, 8, 16)])
def test_x4_upsample_bands_cover_every_output_row_once_and_rolling_rows_hold_the_taps(H, W, LB):
    """Integer model of the kernel's row logic (the arithmetic itself is held bit for bit to the two-stage form on the GPU): every output
    row of the (4H) map is written by exactly one (band, sub-group) share; the two source rows the rolling registers hold when a row is
    written are that row's taps, both inside the band's LDS image; a share forms at most rows / 2 + 2 new horizontal rows (the direct form:
    2 per output row)."""
    Hl, Ho = 2 * H, 4 * H
    Wo = 4 * W
    w4 = Wo // 4
    nsub = 256 // w4 if w4 < 256 else 1
    ry = np.float32(Hl - 1) / np.float32(Ho - 1)
    bands = (Hl + LB - 1) // LB
    written = np.zeros(Ho, dtype=int)
    for band in range(bands):
        ya = band * LB
        yb = min(ya + LB, Hl - 1)
        # yo_first / yo_end exactly as the kernel finds them
        yo_first = int(np.ceil(np.float32(ya) / ry))
        while yo_first > 0 and int(ry * np.float32(yo_first - 1)) >= ya:
            yo_first -= 1
        while int(ry * np.float32(yo_first)) < ya:
            yo_first += 1
        yo_end = yo_first
        while yo_end < Ho and int(ry * np.float32(yo_end)) < ya + LB:
            yo_end += 1
        per = (yo_end - yo_first + nsub - 1) // nsub
        for sub in range(nsub):
            ys = yo_first + sub * per
            ye = min(ys + per, yo_end)
            c0 = c1 = -1
            formed = 0
            for yo in range(ys, ye):
                y0, y1 = _src_tap(ry, yo, Hl)
                if y0 != c0:
                    if y0 != c1:
                        formed += 1
                    c0 = y0
                if y1 != c1:
                    if y1 != c0:
                        formed += 1
                    c1 = y1
                assert (c0, c1) == (y0, y1)
                assert ya <= y0 <= yb and ya <= y1 <= yb          # rows of the band's Lr image
                written[yo] += 1
            if ye > ys:
                assert formed <= (ye - ys) // 2 + 2
    assert (written == 1).all()


# ---- csrc/corr.hip: tile decomposition, ownership of the label-plane pixels and of the gram records, LDS pitch -------------------------
def _corr_tiles(H, W, gram):
    """The kernel's geometry (corr.hip: CORR_TR = 2 rows x 16 columns per wave tile; with the gram a halo row below and a 16th halo
    column: 15 owned columns).  Yields (y0, x0) of every tile of one image."""
    tr, xs = 2, (15 if gram else 16)
    tiles_y, tiles_x = (H + tr - 1) // tr, ((W + 14) // 15 if gram else (W + 15) // 16)
    for ty in range(tiles_y):
        for tx in range(tiles_x):
            yield 1 + tr * ty, 1 + xs * tx


@pytest.mark.parametrize("H,W", [(120, 120), (24, 24), (30, 17), (5, 3), (9, 31), (4, 15), (1, 1), (7, 16)])
@pytest.mark.parametrize("gram", [True, False])
def test_corr_kernel_tiles_cover_every_plane_pixel_and_gram_record(H, W, gram):
    """Restates corr_planes_kernel's store conditions on integers.  Label planes: every interior pixel (1..H, 1..W of the padded map) is
    written, nothing else, and a pixel is written twice only as a tile's 16th (halo) column = the next tile's first.  Gram: every one of the
    five records of every interior pixel is written exactly once, and every fragment the tile reads lies inside the padded map."""
    HP, WP = H + 2, W + 2
    planes = np.zeros((HP, WP), dtype=np.int32)
    recs = np.zeros((H, W, 5), dtype=np.int32)
    for y0, x0 in _corr_tiles(H, W, gram):
        rows = 3 if gram else 2
        for r in range(rows):                                   # loads: rows / columns clamped into the padded map (frag_offs)
            y = min(y0 + r, HP - 1)
            for c in range(16):
                x = min(x0 + c, WP - 1)
                assert 0 <= y < HP and 0 <= x < WP
        for r in range(2):                                      # label planes: lane (c, kg) holds pixels xs .. xs + 3 of row y0 + r
            y = y0 + r
            if y > H:
                continue
            for kg in range(4):
                xs = x0 + 4 * kg
                for e in range(4):
                    if xs + e <= W:                             # (vector store when xs + 3 <= W, masked scalars otherwise: same pixels)
                        planes[y, xs + e] += 1
        if gram:
            for r in range(2):
                y = y0 + r
                if y > H:
                    continue
                for kg in range(4):
                    for e in range(4):
                        m = 4 * kg + e
                        for c in range(16):                     # the lane's column n = c of the 16 x 16 product block
                            if m <= 14 and x0 + m <= W:
                                if c == m:
                                    recs[y - 1, x0 + m - 1, 0] += 1; recs[y - 1, x0 + m - 1, 2] += 1
                                if c == m + 1:
                                    recs[y - 1, x0 + m - 1, 1] += 1; recs[y - 1, x0 + m - 1, 3] += 1
                            if c == m - 1 and c <= 14 and x0 + c <= W:
                                recs[y - 1, x0 + c - 1, 4] += 1
    interior = np.zeros((HP, WP), dtype=bool)
    interior[1:H + 1, 1:W + 1] = True
    assert (planes[interior] >= 1).all() and (planes[~interior] == 0).all()
    if gram:
        assert planes.max() <= 2                                # the halo column only
        twice = np.argwhere(planes == 2)
        assert all((x - 1) % 15 == 0 and x > 1 for _, x in twice), twice[:5]      # = column 0 of the next tile
        assert (recs == 1).all(), np.argwhere(recs != 1)[:5]
    else:
        assert planes.max() == 1


def test_corr_kernel_text_tile_pitch_is_bank_conflict_free_and_fits_the_lds():
    """T in LDS at a pitch of 1024 + 16 bytes: the 16 rows one ds_read_b128 fragment read touches per k-group land on 16 different 16-byte
    bank groups of a 256-byte LDS line (and on 8 different ones of a 128-byte line per 8 lanes); 157 rows are the most that fit 160 KB."""
    pitch = 512 * 2 + 16
    for kg in range(4):
        for ks in range(16):
            offs = [r * pitch + ks * 64 + kg * 16 for r in range(16)]
            assert len({(o % 256) // 16 for o in offs}) == 16
            for half in (offs[:8], offs[8:]):
                assert len({(o % 128) // 16 for o in half}) == 8
    assert 157 * pitch <= 160 * 1024 < 158 * pitch
