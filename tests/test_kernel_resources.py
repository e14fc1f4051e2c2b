"""Register / LDS budgets of the hot kernels, read from the compiler's resource remarks (hipcc
cross-compiles gfx950 without a GPU).  The budgets are what the measured occupancy depends on:
the plain column-FFT pass at N = 512 must stay within 128 VGPRs (two 512-thread workgroups per
CU; at 156 VGPRs r2c went from 1.35 to 1.49 ms), the tile kernels must not spill, and the CIC
tile region must leave room for four workgroups per CU."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'pmesh_amd', 'csrc')
HIPCC = '/opt/rocm/bin/hipcc'


_TABLES = {}


def resources(source):
    """resource table of one source file; the first call compiles BOTH files of this module side by side
    (a minute each: the suite's longest test otherwise)"""
    if not _TABLES:
        from concurrent.futures import ThreadPoolExecutor
        names = ['pmx_colfft.hip', 'pmx_binned.hip']
        with ThreadPoolExecutor(len(names)) as pool:
            for name, table in zip(names, pool.map(_compile_resources, names)):
                _TABLES[name] = table
    if source not in _TABLES:
        _TABLES[source] = _compile_resources(source)
    return _TABLES[source]


def _compile_resources(source):
    cmd = [HIPCC, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-fno-fast-math',
           '-I' + os.path.join(ROOT, 'include'), '-c', os.path.join(CSRC, source), '-o', os.devnull,
           '-Rpass-analysis=kernel-resource-usage']
    out = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    table, name = {}, None
    for line in out.stderr.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            name = m.group(1)
            table[name] = {}
            continue
        m = re.search(r'remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|LDS Size \[bytes/block\]|Occupancy \[waves/SIMD\]): (\d+)', line)
        if m and name:
            table[name][m.group(1).split(' ')[0]] = int(m.group(2))
    return table


pytestmark = pytest.mark.skipif(not os.path.exists(HIPCC), reason='hipcc not installed')


def test_column_fft_budgets():
    t = resources('pmx_colfft.hip')
    # colfft_kernel<T, LOGN, INV, APPLY, RB, REMAP>: plain passes (no transfer, no chunk remap)
    plain = {k: v for k, v in t.items() if 'colfft_kernelId' in k and 'ELb0ELi128ELb0E' in k}
    assert len(plain) >= 10
    for k, v in plain.items():
        assert v['ScratchSize'] == 0, k
    n512 = [v for k, v in plain.items() if 'Li9E' in k]
    assert len(n512) == 2 and all(v['VGPRs'] <= 128 for v in n512), n512
    # float: 512 threads per workgroup at N = 512 as well (16 lines per thread): <= 128 VGPRs
    f512 = [v for k, v in t.items() if 'colfft_kernelIfLi9E' in k]
    assert len(f512) >= 4 and all(v['VGPRs'] <= 128 and v['ScratchSize'] == 0 for v in f512), f512
    # the chunk passes of the pipelined transposes (REMAP), with and without the fused transfer
    chunk = [v for k, v in t.items() if 'colfft_kernelIdLi9E' in k and k.split('ELi128E')[1].startswith('Lb1E')]
    assert len(chunk) == 4 and all(v['VGPRs'] <= 128 and v['ScratchSize'] == 0 for v in chunk), chunk
    for k, v in t.items():
        if 'rowfft_kernel' in k:
            # the 2048-point variants (64-byte tile rows) run one workgroup per CU by their LDS
            # tile anyway: only spills matter there
            # (and the 3 * 2^k variants, whose odd tile shapes the compiler unrolls differently:
            # measured, not budgeted)
            tail = k.split('rowfft_kernel')[1]
            limit = 256 if (tail.count('ELi64E') or any('Li%dE' % c in tail for c in (22, 23, 24, 38, 39))) else 128
            # the forward pass that gathers the staged halos of a paint (HALO = true: the last template argument but one;
            # the last, SEG, is the split layout of the pencil transposes): its tiles leave room for three 256-thread
            # workgroups per CU, i.e. three waves per SIMD: 168 registers
            if re.search(r'ELb0ELi128ELb1ELb0EEE', k):
                limit = 168
            # [r6] float rows of 1024 / 2048 reals are 1024-thread workgroups whose tile is under 80 KB: held to 64
            # registers two of them share a CU (row pass of 1024 reals 3.8 -> 4.9 TB/s); the forms that read the split
            # layout of a pencil transpose park up to 18 words per lane for it (measured: still ahead), the 2048 forward
            # pass two
            spill = 0
            if re.match(r'I[f]Li(9|10)E', tail) and not re.search(r'ELb1ELb0EEE', k):
                limit = 64
                spill = 72 if re.search(r'If\w*Li(9|10)ELb1ELi\d+ELb0ELb1EEE', k) else (8 if 'Li10E' in tail else 0)
            assert v['ScratchSize'] <= spill and v['VGPRs'] <= limit, (k, v)
        if 'colfft_kernel' in k and 'Li11E' in k:
            assert v['ScratchSize'] == 0, (k, v)


def test_tile_kernel_budgets():
    t = resources('pmx_binned.hip')
    tiles = {k: v for k, v in t.items() if 'paint_tile_kernel' in k or 'readout_tile_kernel' in k
             or 'paint_tile32_kernel' in k or 'readout_tile_lean_kernel' in k}
    assert len(tiles) >= 16
    for k, v in tiles.items():
        if 'paint_tile32_kernel' in k:
            # [r5] a few forms (blocks that are not whole meshes, the tile-ordered copy with signed contributions) park up
            # to five kernel-lifetime words per lane — addresses and reciprocals computed once, the carried face — in
            # scratch OUTSIDE the deposit loops (read from the listing: stores at the kernel's entry, reloads at tile
            # boundaries); the forms of the one-rank default path (index list, whole mesh) must not
            assert v['ScratchSize'] <= 24, (k, v)
            if 'ELb0ELb0ELb1E' in k or 'ELb0ELb1ELb1E' in k:
                assert v['ScratchSize'] == 0, (k, v)
        elif 'readout_tile_lean_kernelILi7EdLi768' in k:
            # [r5] PCS on double canvases, 768 threads, an 81 KB region: TWO workgroups per CU need six waves per SIMD,
            # <= 80 VGPRs.  The form for blocks of any shape came to 83 and ran ONE workgroup per CU (2.62 ms against
            # 1.75 at 512^3: every pencil rank of config 5); held to 80 by its launch bound it parks one 8-byte address
            # per lane in scratch, stored at the kernel's entry and reloaded once per TILE (read from the listing:
            # outside the gather loops) — 1.91 ms
            assert v['VGPRs'] <= 80 and v['Occupancy'] >= 6, (k, v)
            assert v['ScratchSize'] <= (16 if 'ELb0EEEv' in k else 0), (k, v)
        elif 'readout_tile_kernelILi7EdLi768' in k:
            # the same budget for the forms behind the tile-ordered copy (rows in random order) and the exact arithmetic:
            # 83-85 VGPRs unbounded, one workgroup per CU; held to 80 they spill up to ten words per lane and are still
            # 12 % faster (512^3 PCS readout of shuffled rows 7.02 -> 6.16 ms, scripts/r05/readout768_ab.sh)
            assert v['VGPRs'] <= 80 and v['Occupancy'] >= 6 and v['ScratchSize'] <= 48, (k, v)
        else:
            assert v['ScratchSize'] == 0, k
        assert v['VGPRs'] <= 128, (k, v)
    # CIC (kind 5), double canvas: 40 KB regions -> four workgroups per CU
    for k, v in tiles.items():
        if 'ILi5Ed' in k:
            assert v['LDS'] <= 40960 and v['Occupancy'] >= 4, (k, v)
    # TSC (kind 6) through the index list, fixed-point regions (MODE 1; double canvases — float ones take the 32-bit
    # regions below): 49 KB regions allow three workgroups of 512 threads per CU only if six waves per SIMD fit the
    # registers (80 VGPRs); at 84-90 it ran two (paint 2.23 instead of 1.85 ms on config 3).  One kernel per form of
    # the deposit loop (whole mesh or not x element size of the positions)
    tsc = [v for k, v in tiles.items() if 'paint_tile_kernelILi6Ed' in k and 'ELb0ELi1E' in k]
    assert len(tsc) == 4 and all(v['VGPRs'] <= 80 and v['Occupancy'] >= 6 for v in tsc), tsc
    # [r6] PCS (kind 7) through the index list on fixed-point regions: the kernel holds both forms of its deposit loop
    # (one lane / four lanes per particle, tile_deposit_quadz); 80 KB regions allow two workgroups of 512 threads per CU,
    # i.e. four waves per SIMD, if 128 VGPRs suffice (the whole-file build came to 129 unbounded: ONE workgroup per CU)
    pcs = [v for k, v in tiles.items() if 'paint_tile_kernelILi7Ed' in k and 'ELb0ELi1E' in k]
    assert len(pcs) == 4 and all(v['VGPRs'] <= 128 and v['Occupancy'] >= 4 and v['ScratchSize'] == 0 for v in pcs), pcs
    # [r5] the 32-bit regions of float canvases: TSC 45 KB (rows of 64 cells) x three workgroups of 512 threads per CU,
    # i.e. six waves per SIMD; PCS four waves per SIMD
    t32 = {k: v for k, v in tiles.items() if 'paint_tile32_kernel' in k}
    assert len(t32) >= 16
    for k, v in t32.items():
        if 'ILi6E' in k:
            assert v['VGPRs'] <= 80 and v['Occupancy'] >= 6 and v['LDS'] <= 49152, (k, v)
        else:
            assert v['VGPRs'] <= 128 and v['Occupancy'] >= 4, (k, v)
    for k, v in t.items():
        if 'bin_count_kernel' in k:
            assert v['ScratchSize'] == 0, k
    # the block form of the single-pass rebuild: a real loop (its first, fully unrolled version had 122
    # VGPRs and 80 KB of code and was no faster than the chunk form), five workgroups per CU by its LDS
    blocks = {k: v for k, v in t.items() if 'bin_block_kernel' in k}
    assert len(blocks) == 8
    for k, v in blocks.items():
        assert v['ScratchSize'] == 0 and v['VGPRs'] <= 96 and v['LDS'] <= 32768, (k, v)
    # [r5] its lean form for dense rows: four rows per lane and trip plus the next trip's four in flight (96 VGPRs for
    # 24-byte rows: five workgroups of 256 threads per CU; measured 0.75 against 1.0 ms of the form above)
    lean = {k: v for k, v in t.items() if 'bin_lean_kernel' in k}
    assert len(lean) == 16
    for k, v in lean.items():
        # [r6] the forms for positions in float (PE = 4) are held to six waves per SIMD — 80 registers, where they took 83
        # and ran five; up to three words per lane parked for it (measured: 0.61 -> 0.56 ms on config 3's rows)
        f4 = re.search(r'bin_lean_kernelILi\dELi4E', k) is not None
        assert v['ScratchSize'] <= (12 if f4 else 0) and v['VGPRs'] <= (80 if f4 else 96) and v['LDS'] <= 20480, (k, v)
        assert not f4 or v['Occupancy'] >= 6, (k, v)
