"""Counter-based RNG spec shared by the oracle and the HIP kernels (numpy restatement).

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference draws its noise with
torch's CPU mt19937 generator (`langevin.py:63,106`, `hmc.py:100,111`, `jump.py:205,225`,
`imh.py:221,229`); an in-kernel generator cannot reproduce that stream, so the build
defines its own *native* stream and this file is its executable specification:

  Philox4x32-10 (Salmon et al., SC'11; constants and known-answer vectors from the
  Random123 distribution's `kat_vectors`; `rounds=7` gives the opt-in Philox4x32-7 stream of
  NfmcRng.rounds), keyed by the 64-bit seed, with counter

      (c0, c1, c2, c3) = (global chain id, transition index, coordinate block, stream tag)

  so a draw is a pure function of (seed, chain, step, coordinate): independent of the
  thread layout and of how chains are sharded over GPUs.

Streams (c3):
  0  proposal noise   MALA epsilon / HMC momentum; block b yields coords 4b..4b+3
  1  accept uniform   one Philox call per 4 transitions: counter (chain, step>>2, 0, 1),
                      word `step & 3`
  2  flow latent      z ~ N(0, I) for Flow.sample; block b yields coords 4b..4b+3
  3  jump uniform     counter (chain, step, 0, 3), word 0

Uniform / normal transforms (all in fp32 unless stated):
  uniform   u  = (2*(r >> 9) + 1) * 2^-24            in (0, 1), exact
  normal    u1 = fl(fl(r_a) * 2^-32 + 2^-33),  u2 = fl(r_b) * 2^-32   (fl = u32 -> f32 RNE)
            R  = sqrt(-2 ln u1);  z_a = R cos(2 pi u2);  z_b = R sin(2 pi u2)
            pairs (r0, r1) -> coords (4b, 4b+1), (r2, r3) -> (4b+2, 4b+3)
  The oracle evaluates ln/sqrt/cos/sin in fp64 and rounds once; the kernel uses the
  gfx950 hardware transcendentals (v_log_f32, v_sqrt_f32, v_cos_f32, v_sin_f32), so
  native-mode noise agrees to ~1e-6 absolute, not bitwise.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)

TAG_NOISE = 0
TAG_ACCEPT = 1
TAG_LATENT = 2
TAG_JUMP = 3

_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1, rounds=10):
    """Vectorised Philox4x32-`rounds` (10: the library's stream; 7: the opt-in stream, NfmcRng.rounds).  All inputs
    broadcastable uint32 arrays; returns 4 uint32 arrays."""
    c0, c1, c2, c3 = np.broadcast_arrays(*(np.asarray(v, dtype=np.uint32) for v in (c0, c1, c2, c3)))
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over='ignore'):
        for rnd in range(rounds):
            p0 = PHILOX_M0 * c0.astype(np.uint64)
            p1 = PHILOX_M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            if rnd != rounds - 1:
                k0 = np.uint32(k0 + PHILOX_W0)
                k1 = np.uint32(k1 + PHILOX_W1)
    return c0, c1, c2, c3


def _seed_key(seed):
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    return np.uint32(seed & 0xFFFFFFFF), np.uint32(seed >> 32)


def u32_to_uniform(r):
    """(0,1) uniform with 23 random bits, exact in fp32."""
    r = np.asarray(r, dtype=np.uint32)
    return ((2 * (r >> np.uint32(9)).astype(np.int64) + 1).astype(np.float32) * np.float32(2.0 ** -24)).astype(np.float32)


def box_muller(ra, rb):
    """Two standard normals from two uint32 words (spec in module docstring)."""
    fa = np.asarray(ra, dtype=np.uint32).astype(np.float32)
    fb = np.asarray(rb, dtype=np.uint32).astype(np.float32)
    u1 = (fa * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)).astype(np.float32)
    u2 = (fb * np.float32(2.0 ** -32)).astype(np.float32)
    rad = np.sqrt(-2.0 * np.log(u1.astype(np.float64)))
    ang = 2.0 * np.pi * u2.astype(np.float64)
    return (rad * np.cos(ang)).astype(np.float32), (rad * np.sin(ang)).astype(np.float32)


def normal_field(seed, chain_ids, step, d, tag, rounds=10):
    """(len(chain_ids), d) fp32 standard normals of one transition for stream `tag`."""
    k0, k1 = _seed_key(seed)
    chain_ids = np.asarray(chain_ids, dtype=np.uint32)
    nblk = (d + 3) // 4
    blocks = np.arange(nblk, dtype=np.uint32)
    r0, r1, r2, r3 = philox4x32_10(chain_ids[:, None], np.uint32(step), blocks[None, :], np.uint32(tag), k0, k1, rounds)
    z0, z1 = box_muller(r0, r1)
    z2, z3 = box_muller(r2, r3)
    out = np.stack([z0, z1, z2, z3], axis=-1).reshape(len(chain_ids), nblk * 4)
    return np.ascontiguousarray(out[:, :d])


def accept_uniform(seed, chain_ids, step, rounds=10):
    """(len(chain_ids),) fp32 uniforms for the Metropolis test of transition `step` (stream 1)."""
    k0, k1 = _seed_key(seed)
    chain_ids = np.asarray(chain_ids, dtype=np.uint32)
    r = philox4x32_10(chain_ids, np.uint32(step >> 2), np.uint32(0), np.uint32(TAG_ACCEPT), k0, k1, rounds)
    return u32_to_uniform(r[step & 3])


def jump_uniform(seed, chain_ids, step, rounds=10):
    """(len(chain_ids),) fp32 uniforms for the flow-proposal MH test of transition `step` (stream 3)."""
    k0, k1 = _seed_key(seed)
    chain_ids = np.asarray(chain_ids, dtype=np.uint32)
    r = philox4x32_10(chain_ids, np.uint32(step), np.uint32(0), np.uint32(TAG_JUMP), k0, k1, rounds)
    return u32_to_uniform(r[0])
