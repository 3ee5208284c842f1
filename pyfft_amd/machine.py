"""The planner's model of the memory system, derived from mifft_device_props instead of literals (round 4).

Everything `FFTPlan._select_strategy` sizes -- the ring of the persistent launches, the chunks of the pipelined chain, the
thresholds of the plain chain and of the write-through store rule -- is a fraction of the LAST-LEVEL CACHE in front of HBM (the
256 MiB Infinity Cache of an MI355X; `llc_bytes`, read from the HSA agent by mifft_device_props_get), and the grids of the
persistent kernels are multiples of the compute-unit count.  The fractions are the measured ones (docs/strategies.md); on the
full part they give back the constants of rounds 1-3: 224 MiB ring, 64 MiB chunks, 128 MiB slabs, 256 MiB chain threshold,
128 MiB write-through threshold.  A partition of the part (CPX / NPS4: 32 CUs, one XCD, a slice of the cache) gets
proportionally smaller figures and loses the strategies that need the whole machine (XCD-cooperative kernels) or a ring of at
least four transforms.  The reference has no counterpart: it reads only block / grid / shared-memory limits (pyfft/cuda.py:72-83).

Round 5: the fractions and the ring rule themselves are data -- `cache_fractions` / `ring_rule` of pyfft_amd/tuning_gfx950.json
(pyfft_amd/tuning.py); this module only applies them to the device's figures.
"""
from . import tuning as _tuning


class Machine(object):
    __slots__ = ("compute_units", "num_xcc", "l2_bytes", "llc_bytes", "tuning")

    def __init__(self, compute_units, num_xcc, l2_bytes, llc_bytes, tuning=None):
        self.compute_units = int(compute_units)
        self.num_xcc = max(1, int(num_xcc))
        self.l2_bytes = int(l2_bytes)
        self.llc_bytes = max(0, int(llc_bytes))
        self.tuning = tuning if tuning is not None else _tuning.default()

    @property
    def MIN_RING_SLOTS(self):
        """A persistent two-pass launch needs producers a few transforms ahead of the consumers."""
        return int(self.tuning.ring_rule["min_ring_slots"])

    @classmethod
    def from_props(cls, props, tuning=None):
        return cls(props.compute_units, props.num_xcc, props.l2_bytes, props.llc_bytes, tuning)

    # ---- fractions of the last-level cache -------------------------------------------------------------------------
    @property
    def ring_bytes(self):
        """Largest intermediate ring that still lives in the cache next to the streams (7/8 of it: 224 of 256 MiB measured
        best, 18 or 36 slots of 8 MiB: 36 % against 37.3 %)."""
        return self.tuning.fraction("ring", self.llc_bytes)

    @property
    def pipeline_chunk_bytes(self):
        """Chunk of the pipelined chain: the chunk's input side + its intermediate + the other stream's chunk share the cache."""
        return self.tuning.fraction("pipeline_chunk", self.llc_bytes)

    @property
    def slab_bytes(self):
        """Slabs of the leading passes of a 3-D transform bigger than the cache (C4: 24.7 % at a quarter, 25.5 % at half)."""
        return self.tuning.fraction("slab", self.llc_bytes)

    @property
    def chain_max_bytes(self):
        """Per side: up to the cache size one launch per pass over the whole batch beats chunks and persistent launches."""
        return self.tuning.fraction("chain_max", self.llc_bytes)

    @property
    def write_through_max_bytes(self):
        """Per side: below this every launch stores write-through (the end-of-kernel write-back of a small launch runs alone)."""
        return self.tuning.fraction("write_through_max", self.llc_bytes)

    @property
    def stream_hint_item_bytes(self):
        """Non-temporal first-load / last-store hints only while a transform's intermediate can stay in the cache."""
        return self.tuning.fraction("stream_hint_item", self.llc_bytes)

    # ---- persistent launches -----------------------------------------------------------------------------------------
    def fused_geometry(self, item_bytes, tiles0, groups_per_cu, fill_cache=False, min_slots=None):
        """(lag, ring, grid) of a persistent two-pass launch, or None when the cache holds no useful ring.
        lag: the producers stay 1.75 work-group waves of first-pass tiles ahead of the consumers (C2: 14 transforms of 64 tiles
        on 512 work-groups; 2^19 has 32 tiles per transform and ran 4 points low on 14: profiles/r04_a_fused_sweep.log);
        ring = 2 * lag, capped by the cache -- then the consumers follow by 4/7 of the ring (measured on the 32 MiB transforms)."""
        rr = self.tuning.ring_rule
        grid = groups_per_cu * self.compute_units
        slots = self.ring_bytes // max(1, item_bytes)
        if slots < (min_slots or self.MIN_RING_SLOTS) or grid < 1:
            return None
        wn, wd = rr["lag_waves"]
        lag = max(2, -(-wn * grid // (wd * max(1, tiles0))))
        ring = 2 * lag
        if ring > slots or fill_cache:
            # fill_cache: transforms of many small tiles (the 128^3 cubes: 512 tiles of 32-64 KiB) want the whole ring whatever
            # the tile count says -- fp64 128^3: lag 2 / ring 4 0.366, 4 / 7 0.388 (profiles/r04_b_cube_sweep.log); the smaller
            # 3-D shapes up to 56 slots: 64^3 fp32 14 slots 0.246, 28 0.395, 56 0.418, 112 0.414; (64, 128, 128) 14 0.366, 28 0.417
            # (profiles/r04_z_pair_small_axes_rings.log)
            ring = min(slots, int(rr["fill_cache_ring_slots"] if fill_cache else rr["capped_ring_slots"]))
            lag = max(1, rr["capped_lag"][0] * ring // rr["capped_lag"][1])
        return lag, ring, grid

    @property
    def xcd_cooperative(self):
        """The XCD-cooperative kernels (xcd2; per-XCD work lists) assume 8 XCDs of 32 CUs."""
        return self.num_xcc == 8 and self.compute_units == 32 * self.num_xcc

    def __repr__(self):
        return "Machine(cus=%d, xcc=%d, l2=%d KiB, llc=%d MiB)" % (self.compute_units, self.num_xcc, self.l2_bytes >> 10, self.llc_bytes >> 20)
