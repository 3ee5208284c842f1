"""Axis -> pass decomposition for gfx950 (host side; counterpart of pyfft/kernel.py and
pyfft/kernel_helpers.py in the reference).

The reference factors a long or strided axis in base 128 (kernel_helpers.py:67-122) and an
LDS-resident axis by a fixed table up to n = 2048 (kernel_helpers.py:10-65) because of 16 KiB
of shared memory and 512-thread blocks.  MI355X has 160 KiB of LDS per CU and 1024-thread
work-groups, so the factorisation here is different: a contiguous axis up to 16384 points
(32768 interleaved fp32; half of that in fp64) is one ROW launch, a whole 2-D/3-D shape of up to
16384 points (more for the common shapes, see mifft_nd_shape_supported) is one ND launch, a
strided axis up to 1024 points is one COL launch, and longer axes are split into the fewest
near-equal COL passes (every pass is a full HBM round trip, so fewer is better).
The pass algebra itself is the reference's (kernel.mako:805-1047):

    view in as [outer][R][M][S], out as [outer][M][R][S]
    out[l][q][j] = w(R*M)^(l*q) * sum_r in[r][l][j] * w(R)^(r*q)
"""

from . import _native as N

X_DIRECTION, Y_DIRECTION, Z_DIRECTION = 0, 1, 2


def log2(n):
    """Integer log2 (kernel_helpers.py:2-8)."""
    n = int(n)
    r = 0
    while n > 1:
        n >>= 1
        r += 1
    return r


def is_pow2(n):
    return n >= 1 and (n & (n - 1)) == 0


_support_cache = {}


def _max_len(kind, precision, variant=0):
    """Largest L the library has a compiled kernel for (Function.isExecutable counterpart,
    cuda.py:48-49)."""
    key = (kind, precision, variant)
    if key not in _support_cache:
        best = 0
        L = 2
        while L <= (1 << 16):
            if N.lib.mifft_pass_supported(kind, precision, L, 0) == 0 or \
                    (variant and N.lib.mifft_pass_supported(kind, precision, L, variant) == 0):
                best = L
            L *= 2
        _support_cache[key] = best
    return _support_cache[key]


def row_max(precision, interleaved=False):
    """Longest contiguous axis one ROW launch takes; the longest rows exist for interleaved data only."""
    return _max_len(N.PASS_ROW, precision, N.VARIANT_INTERLEAVED_ONLY if interleaved else 0)


def col_max(precision):
    return _max_len(N.PASS_COL, precision)


def split_radices(n, max_radix):
    """Fewest near-equal power-of-two factors of n, each <= max_radix, largest first."""
    bits = log2(n)
    maxbits = log2(max_radix)
    npass = max(1, -(-bits // maxbits))
    base, extra = divmod(bits, npass)
    return [1 << (base + (1 if i < extra else 0)) for i in range(npass)]


class PassSpec(object):
    """Shape of one launch, independent of batch/direction (those are filled in by the plan).
    Counterpart of a compiled LocalFFTKernel / GlobalFFTKernel object (kernel.py:124-283)."""

    __slots__ = ("kind", "axis", "n", "L", "M", "S", "outer_per_batch", "outer_stride",
                 "in_place_possible", "curr_n", "pair_with_next")

    def __init__(self, kind, axis, n, L, M, S, outer_per_batch, outer_stride, in_place_possible):
        self.pair_with_next = False      # this pass and the next one are run by ONE launch (csrc/fft_pair.hpp)
        self.kind = kind
        self.axis = axis
        self.n = n
        self.L = L
        self.M = M
        self.S = S
        self.outer_per_batch = outer_per_batch
        self.outer_stride = outer_stride
        self.in_place_possible = in_place_possible
        self.curr_n = L * M

    def __repr__(self):
        if self.kind == N.PASS_ND:
            return "nd(x=%d,y=%d,z=%d)*" % (self.L, self.M, self.S)
        name = "row" if self.kind == N.PASS_ROW else "col"
        return "%s(L=%d,M=%d,S=%d)%s%s" % (name, self.L, self.M, self.S, "*" if self.in_place_possible else "",
                                            " +" if self.pair_with_next else "")


def col_chain(axis, n, radix_init, outer_per_batch, precision, interleaved=True):
    """Chain of COL passes for an axis of length n whose faster axes multiply to radix_init
    (GlobalFFTKernel.createChain, kernel.py:259-283; strides kernel.py:196-204).  Only the last
    pass (M == 1) can run in place (cf. kernel.py:238-241)."""
    cmax = col_max(precision)
    # fp64 L = 2048 exists as a fast kernel for interleaved data with 8 whole columns per tile (csrc/fft_colx.hpp: S == 1 or
    # S >= 8); anything else would take its 4-column fallback (64-byte segments), where the 1024-point factorisation is faster
    if precision == N.F64 and cmax > 1024 and (not interleaved or 1 < radix_init < 8):
        cmax = 1024
    radices = split_radices(n, cmax)
    chain = []
    S = radix_init
    curr_n = n
    for R in radices:
        M = curr_n // R
        chain.append(PassSpec(N.PASS_COL, axis, n, R, M, S, outer_per_batch, n * radix_init, M == 1))
        S *= R
        curr_n //= R
    return chain


def pair_chain(x, y, z, r0):
    """The four passes of a (z, y, x) transform with the y axis factored r0 * r1 (the chain's own factorisation of a long axis,
    kernel.py:259-283), marked as the two pass pairs csrc/fft_pair.hpp runs: (ROW x, COL y r0) and (COL y r1, COL z)."""
    r1 = y // r0
    chain = [PassSpec(N.PASS_ROW, X_DIRECTION, x, x, 1, 1, y * z, x, True),
             PassSpec(N.PASS_COL, Y_DIRECTION, y, r0, r1, x, z, y * x, False),
             PassSpec(N.PASS_COL, Y_DIRECTION, y, r1, 1, x * r0, z, y * x, True),
             PassSpec(N.PASS_COL, Z_DIRECTION, z, z, 1, x * y, 1, x * y * z, True)]
    chain[0].pair_with_next = True
    chain[2].pair_with_next = True
    return chain


def build_chain(x, y, z, precision, interleaved=False):
    """Kernel chain for a (z, y, x) array with x contiguous (FFTPlan._generateKernelCode and
    _fft1D, plan.py:111-171): X passes, then the Y chain, then the Z chain.  Axes of length 1 are
    skipped (plan.py:149,160,164).  `interleaved`: the user's buffers are interleaved complex (some
    kernels do not take split planes)."""
    chain = []
    variant = N.VARIANT_INTERLEAVED_ONLY if interleaved else 0

    def nd_ok(xx, yy, zz):
        return N.lib.mifft_nd_shape_supported(precision, xx, yy, zz, variant) == 0

    # small 2-D / 3-D transforms: every axis inside LDS, one launch, one HBM round trip (csrc/fft_nd.hpp, fft_nd2.hpp)
    ndims = (x > 1) + (y > 1) + (z > 1)
    # (split-complex plans: a few shapes beyond the run-time-shaped kernel's tile exist for planes on both sides of ONE launch -- the tiled
    # fixed-shape kernel with one tile per parent, second batch of round 4: (32, 32, 32) fp32 0.31 as two passes)
    if ndims >= 2 and x * y * z >= 4 and (nd_ok(x, y, z) or (not interleaved and
                                                             N.lib.mifft_nd_shape_supported(precision, x, y, z, N.VARIANT_SPLIT_ONLY) == 0)):
        return [PassSpec(N.PASS_ND, X_DIRECTION, x * y * z, x, y, z, 1, x * y * z, True)]
    # 3-D shapes the library has pass-pair kernels for: two launches of two passes each -- (ROW x, COL y R0) on R0 rows of a plane
    # and (COL y R1, COL z) on 128-byte column segments (csrc/fft_pair.hpp).  256^3: the plane fits no tile, so this replaces one
    # HBM round trip per axis; 128^3: two launches on 32-64 KiB tiles instead of a 128 KiB plane tile + a generic column pass
    if ndims == 3:
        r0 = N.lib.mifft_pair_split(precision, N.INTERLEAVED if interleaved else N.SPLIT, x, y, z)
        if r0 > 0:
            return pair_chain(x, y, z, r0)
    # 3-D shapes too big for one tile but with a small (y, x) plane: x and y together in LDS per plane (the planes
    # are just more batch items), then only z as a strided chain -- two HBM round trips instead of three
    # (split-complex plans: also the planes that exist for plane input only as the tiled fixed-shape kernel -- fp64 (128, 128))
    if ndims == 3 and (nd_ok(x, y, 1) or (not interleaved and N.lib.mifft_nd_shape_supported(precision, x, y, 1, N.VARIANT_SPLIT_ONLY) == 0)):
        return [PassSpec(N.PASS_ND, X_DIRECTION, x * y, x, y, 1, z, x * y, True)] + \
            col_chain(Z_DIRECTION, z, x * y, 1, precision, interleaved)
    # Round 5: ONE pass pair where the chain below would take a third launch (csrc/fft_pair_f32.hip / _f64.hip list the shapes):
    # a 3-D shape with short y and z behind a long x -- the y and z passes as one launch on whole (z, y) planes;
    if ndims == 3 and _pair_kernel(precision, interleaved, 1, x, y, z):
        return yz_pair_chain(x, y, z, precision, interleaved)
    # a y axis too long for one strided pass behind a contiguous x axis -- the row pass and the first y pass as one launch
    if x > 1 and y > col_max(precision) and x <= row_max(precision, interleaved):
        r0 = xy_pair_split(x, y, precision, interleaved)
        if r0:
            return xy_pair_chain(x, y, z, r0, precision, interleaved)
    if x > 1:
        if x <= row_max(precision, interleaved):
            chain.append(PassSpec(N.PASS_ROW, X_DIRECTION, x, x, 1, 1, y * z, x, True))
        else:
            chain.extend(col_chain(X_DIRECTION, x, 1, y * z, precision, interleaved))
    if y > 1:
        if x == 1:
            # degenerate: the y axis is the contiguous one
            sub = build_chain(y, 1, 1, precision, interleaved)
            for p in sub:
                p.axis = Y_DIRECTION
                p.outer_per_batch *= z
            chain.extend(sub)
        else:
            chain.extend(col_chain(Y_DIRECTION, y, x, z, precision, interleaved))
    if z > 1:
        if x * y == 1:
            sub = build_chain(z, 1, 1, precision, interleaved)
            for p in sub:
                p.axis = Z_DIRECTION
            chain.extend(sub)
        else:
            chain.extend(col_chain(Z_DIRECTION, z, x * y, 1, precision, interleaved))
    return chain


class _Unit(object):
    """One launch of a chain as the buffer schedule sees it."""
    __slots__ = ("in_place_possible", "pair_with_next")

    def __init__(self, in_place_possible):
        self.in_place_possible = in_place_possible
        self.pair_with_next = False


def _pair_kernel(precision, interleaved, kind, k0, k1, k2):
    return N.lib.mifft_pair_kernel_supported(precision, N.INTERLEAVED if interleaved else N.SPLIT, kind, int(k0), int(k1), int(k2)) == 0


def xy_pair_split(x, y, precision, interleaved):
    """R0 of a (ROW x, COL y R0) pair kernel for a y axis that is too long for one strided pass -- the remaining R1 = y / R0 points
    must be ONE plain pass -- or 0.  Round 5: (4096, 256) as pair + pass instead of row + col 64 + col 64 (three launches)."""
    cmax = col_max(precision)
    r0 = y // 2
    while r0 >= 2:
        r1 = y // r0
        if r1 <= cmax and N.lib.mifft_pass_supported(N.PASS_COL, precision, r1, 0) == 0 and _pair_kernel(precision, interleaved, 0, x, r0, r1):
            return r0
        r0 //= 2
    return 0


def xy_pair_chain(x, y, z, r0, precision, interleaved):
    """ROW x + COL y (r0) as one launch, the plain COL y (r1) pass, then the z chain (the chain's own factorisation of a long axis,
    kernel.py:259-283, with its first two passes paired: csrc/fft_pair.hpp PairXY)."""
    r1 = y // r0
    chain = [PassSpec(N.PASS_ROW, X_DIRECTION, x, x, 1, 1, y * z, x, True),
             PassSpec(N.PASS_COL, Y_DIRECTION, y, r0, r1, x, z, y * x, False),
             PassSpec(N.PASS_COL, Y_DIRECTION, y, r1, 1, x * r0, z, y * x, True)]
    chain[0].pair_with_next = True
    if z > 1:
        chain.extend(col_chain(Z_DIRECTION, z, x * y, 1, precision, interleaved))
    return chain


def yz_pair_chain(x, y, z, precision, interleaved):
    """X passes, then COL y + COL z as one launch on whole (z, y) planes of 16 adjacent x (csrc/fft_pair.hpp PairYZ with the y axis
    unsplit): a 3-D shape with short y and z behind a long x in two launches instead of three."""
    if x <= row_max(precision, interleaved):
        chain = [PassSpec(N.PASS_ROW, X_DIRECTION, x, x, 1, 1, y * z, x, True)]
    else:
        chain = col_chain(X_DIRECTION, x, 1, y * z, precision, interleaved)
    chain += [PassSpec(N.PASS_COL, Y_DIRECTION, y, y, 1, x, z, y * x, True),
              PassSpec(N.PASS_COL, Z_DIRECTION, z, z, 1, x * y, 1, x * y * z, True)]
    chain[-2].pair_with_next = True
    return chain


def launch_units(chain):
    """[(unit, number of descriptors)]: a pass, or a pass pair (in-place capable only when both passes are)."""
    units, i = [], 0
    while i < len(chain):
        if chain[i].pair_with_next:
            units.append((_Unit(chain[i].in_place_possible and chain[i + 1].in_place_possible), 2))
            i += 2
        else:
            units.append((_Unit(chain[i].in_place_possible), 1))
            i += 1
    return units


def buffer_schedule(chain, is_inplace, via_temp=False):
    """Which buffer every pass reads and writes: returns (temp_needed, [(src, dst), ...]) with 0 = data_in,
    1 = data_out, 2 = the plan's temp buffer.  Contract of FFTPlan._execute (plan.py:194-248): an out-of-place call never
    writes data_in, the result ends in data_out (which aliases data_in for an in-place call, so index 1 is the user's
    buffer in both cases), and a pass that cannot run in place never reads and writes the same buffer.

    The chain alternates between data_out and temp, and the LAST pass must land on data_out: so pass i (of n) writes data_out
    exactly when an even number of buffer switches follows it.  An in-place call starts ON data_out; with an odd number of
    passes one pass has to stay where it is, and that is the first one that can run in place (plan.py:214-221)."""
    if any(getattr(p, "pair_with_next", False) for p in chain):
        # a pass pair is ONE launch: schedule the units, then give both descriptors of a pair the unit's (src, dst)
        units = launch_units(chain)
        temp_needed, usched = buffer_schedule([u for u, _ in units], is_inplace, via_temp)
        sched = []
        for (u, count), sd in zip(units, usched):
            sched.extend([sd] * count)
        return temp_needed, sched
    n = len(chain)
    temp_needed = any(not p.in_place_possible for p in chain)
    start = 1 if is_inplace else 0
    if via_temp and not temp_needed and n >= 2:
        # every pass can run in place: go in -> temp, stay on the temp, last pass temp -> out
        # (used for fp32 split-plane plans, whose temp is interleaved: only two sides touch the planes)
        return True, [(start, 2)] + [(2, 2)] * (n - 2) + [(2, 1)]
    if not temp_needed:
        return False, [(0 if i == 0 else 1, 1) for i in range(n)]
    stays = None
    if is_inplace and n % 2 == 1:
        stays = next((i for i, p in enumerate(chain) if p.in_place_possible), None)
    sched, here = [], start
    for i in range(n):
        if i == stays:
            there = here
        elif is_inplace:
            there = 2 if here == 1 else 1
        else:
            switches_after = n - 1 - i          # every later pass switches buffers
            there = 1 if switches_after % 2 == 0 else 2
        sched.append((here, there))
        here = there
    return temp_needed, sched
