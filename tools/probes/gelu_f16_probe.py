"""can the fused-MLP activations run on packed f16 (v_pk_fma_f16)?   CPU probe, no GPU needed.

    python tools/probes/gelu_f16_probe.py > profiles/r05_gelu_f16_probe.txt

Evaluates  Phi(x) - 1/2 = x P(x^2)  and  GELU'(x) - 1/2 = x Q(x^2)  with every operation rounded to IEEE binary16 (one rounding per fused
multiply-add, like v_pk_fma_f16), over EVERY bf16 value in [-4, 4], against erf-GELU in float64.  Bound to meet (the LUT path's own,
DESIGN 3.4 "GELU in the bf16 kernels"): |delta GELU| <= 1e-3, |delta GELU'| <= 1e-3 absolute.
Forms tried: (a) Horner in t = x^2 / 16 (coefficients O(1)); (b) Horner in the shifted variable t - 1/2 (smaller intermediate values);
(c) the last K Horner steps and the final x * r + 1/2 in f32 (v_fma_mix_f32), the first steps in f16.
"""
import math
import numpy as np

f16 = np.float16


def fma16(a, b, c):                     # one rounding: the products of two binary16 numbers are exact in float64
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f16)


def mul16(a, b):
    return (a.astype(np.float64) * b.astype(np.float64)).astype(f16)


def all_bf16(lo=-4.0, hi=4.0):
    bits = np.arange(0, 1 << 16, dtype=np.uint32) << 16
    v = bits.view(np.float32)
    v = v[np.isfinite(v)]
    return np.unique(v[(v >= lo) & (v <= hi)]).astype(np.float64)


def phi(x):
    return np.array([0.5 * (1.0 + math.erf(t / math.sqrt(2.0))) for t in x])


def dgelu(x):
    return phi(x) + x * np.exp(-0.5 * x * x) / math.sqrt(2 * math.pi)


def fit(fun, deg, shift, n=4001):
    """least squares on Chebyshev nodes of u in [0, 1] (t = u), target fun(x) with x = 4 sqrt(t); polynomial in (t - shift)"""
    k = np.arange(n)
    t = 0.5 - 0.5 * np.cos(np.pi * (k + 0.5) / n)
    x = 4.0 * np.sqrt(t)
    y = fun(x)                      # (f(x) - 1/2) / x
    A = np.vander(t - shift, deg + 1, increasing=False)
    # weight by x: the quantity that matters is x * r
    w = x
    c, *_ = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)
    return c                        # highest degree first


def horner16(c, x64, shift, f32_tail=0):
    """x64: bf16 values as float64.  Returns x * r(t) + 1/2 with the arithmetic described above."""
    xh = x64.astype(f16)
    th = mul16(mul16(xh, xh), np.full_like(xh, 1.0 / 16.0))
    if shift:
        th = (th.astype(np.float64) - shift).astype(f16)
    ch = [np.full_like(xh, f16(v)) for v in c]
    n16 = len(c) - 1 - f32_tail
    r = ch[0]
    for i in range(1, 1 + n16):
        r = fma16(r, th, ch[i])
    if f32_tail:
        r32 = r.astype(np.float32)
        t32 = th.astype(np.float32)
        for i in range(1 + n16, len(c)):
            r32 = (r32.astype(np.float64) * t32 + np.float32(c[i])).astype(np.float32)
        return (x64.astype(np.float32).astype(np.float64) * r32 + 0.5).astype(np.float32).astype(np.float64)
    # final step in f32 from f16 operands (v_fma_mix_f32: x f32, r f16): one rounding, exact enough
    return (x64 * r.astype(np.float64) + 0.5).astype(np.float32).astype(np.float64)


def report(name, val, ref, x):
    e = np.abs(val - ref)
    i = int(np.argmax(e))
    print(f"  {name:<34s} max |delta| {e.max():.3e} at x = {x[i]:+.4f}   (mean {e.mean():.2e})")
    return e.max()


def main():
    x = all_bf16()
    print(f"{len(x)} bf16 values in [-4, 4]; reference erf-GELU in float64; bound 1e-3 on GELU and on GELU'")
    P, D = phi(x), dgelu(x)
    G = x * P
    with np.errstate(divide="ignore", invalid="ignore"):
        fphi = lambda z: np.where(z == 0, 1 / math.sqrt(2 * math.pi), (phi(z) - 0.5) / np.where(z == 0, 1, z))
        fdg = lambda z: np.where(z == 0, 2 / math.sqrt(2 * math.pi), (dgelu(z) - 0.5) / np.where(z == 0, 1, z))
    ok = {}
    for deg_phi, deg_dg in ((6, 7), (7, 8)):
        for shift in (0.0, 0.5):
            for tail in (0, 1, 2, 3):
                print(f"degree {deg_phi} / {deg_dg} in t = x^2/16{' - 1/2' if shift else ''}, last {tail} Horner step(s) in f32:")
                cp, cd = fit(fphi, deg_phi, shift), fit(fdg, deg_dg, shift)
                # f64 evaluation of the same polynomial = the fit's own error
                t = x * x / 16 - shift
                e_fit_g = np.abs(x * (x * np.polyval(cp, t) + 0.5) - G).max()
                e_fit_d = np.abs((x * np.polyval(cd, t) + 0.5) - D).max()
                print(f"  fit alone (float64 arithmetic)      GELU {e_fit_g:.2e}  GELU' {e_fit_d:.2e};  max |coefficient| {np.abs(cp).max():.2f} / {np.abs(cd).max():.2f}")
                ph = horner16(cp, x, shift, tail)
                dg = horner16(cd, x, shift, tail)
                eg = report("GELU = x * Phi_f16", x * ph, G, x)
                report("Phi_f16", ph, P, x)
                ed = report("GELU'_f16", dg, D, x)
                # the error relative to the bf16 rounding the result undergoes right after (half an ulp of bf16 at the result's magnitude)
                ulp = np.maximum(np.abs(G), 2.0 ** -126) * 2.0 ** -9
                print(f"  GELU error / (bf16 half-ulp of GELU)  max {np.max(np.abs(x * ph - G) / ulp):.2f} (|x| >= 0.5: {np.max((np.abs(x * ph - G) / ulp)[np.abs(x) >= 0.5]):.2f})")
                ok[(deg_phi, shift, tail)] = (eg, ed)
    print("summary (GELU, GELU' max abs error; PASS needs both <= 1e-3):")
    for k, (eg, ed) in ok.items():
        print(f"  degree {k[0]}, shift {k[1]}, f32 tail {k[2]}: {eg:.2e} {ed:.2e} {'PASS' if eg <= 1e-3 and ed <= 1e-3 else 'fail'}")


def kernel_form(c, x32, clamp=4.0):
    """Exactly what gelu_h16_* in csrc/common.h executes: v_cvt_pk_f16_f32 (RNE), v_pk_max/min_f16 (+-4), q = x/4, t = fma(q, q, -1/2),
    Horner steps in f16 except the last, which is v_fma_mix_f32 with an f32 constant, then fma(xc, r, 1/2) in f32 from the f16 xc."""
    xh = np.clip(x32.astype(f16), f16(-clamp), f16(clamp))
    q = mul16(xh, np.full_like(xh, 0.25))
    t = fma16(q, q, np.full_like(xh, -0.5))
    r = np.full_like(xh, f16(c[0]))
    for v in c[1:-1]:
        r = fma16(r, t, np.full_like(xh, f16(v)))
    r32 = (r.astype(np.float64) * t.astype(np.float64) + np.float32(c[-1])).astype(np.float32)
    return (xh.astype(np.float64) * r32 + 0.5).astype(np.float32)


def final():
    print()
    print("==== the form that went into csrc/common.h (gelu_h16_phi2 / gelu_h16_dg2): degree 7 / 7 in t = (x/4)^2 - 1/2, last Horner step in f32")
    with np.errstate(divide="ignore", invalid="ignore"):
        fphi = lambda z: np.where(z == 0, 1 / math.sqrt(2 * math.pi), (phi(z) - 0.5) / np.where(z == 0, 1, z))
        fdg = lambda z: np.where(z == 0, 2 / math.sqrt(2 * math.pi), (dgelu(z) - 0.5) / np.where(z == 0, 1, z))
    DEG_PHI = 7
    cp, cd = fit(fphi, DEG_PHI, 0.5), fit(fdg, 7, 0.5)
    # the clamp of the Phi form: the f16 value c in [3.9, 4] whose evaluated Phi(-c) is the smallest non-negative number (the tail of GELU is then
    # x * Phi(-c) with the right sign, and 1 - Phi(c) mirrors it: the polynomial is odd about 1/2 in exact arithmetic and the roundings are sign-symmetric)
    cands = np.arange(3.9, 4.0 + 1e-9, 2.0 ** -9).astype(np.float32)
    lo = kernel_form(cp, -cands, 4.0).astype(np.float64)
    hi = kernel_form(cp, cands, 4.0).astype(np.float64)
    good = np.where((lo >= 0) & (hi <= 1))[0]
    best = good[np.argmin(lo[good])]
    CL = float(cands[best])
    print(f"  Phi clamp {CL!r}: Phi(-c) = {lo[best]:.3e}, 1 - Phi(c) = {1 - hi[best]:.3e} (true Phi(-c) = {phi(np.array([-CL]))[0]:.2e})")
    for name, c in (("MVLT_H16_PHI", cp), ("MVLT_H16_DG", cd)):
        print(f"  {name}: f16 constants " + ", ".join(f"{float(f16(v)):.6e}f" for v in c[:-1]) + f"; f32 tail {np.float32(c[-1]):.9e}f")
    rng = np.random.default_rng(5)
    sets = {"all bf16 values in [-4, 4]": all_bf16().astype(np.float32),
            "2^22 random f32 in [-4.5, 4.5] (the forward's accumulators are f32)": rng.uniform(-4.5, 4.5, 1 << 22).astype(np.float32),
            "2^20 f32 ~ N(0, 1.5^2)": (1.5 * rng.standard_normal(1 << 20)).astype(np.float32),
            "tail: f32 in [4, 12] and [-12, -4]": np.concatenate([np.linspace(4, 12, 4001), -np.linspace(4, 12, 4001)]).astype(np.float32)}
    worst = [0.0, 0.0]
    for name, xs in sets.items():
        x64 = xs.astype(np.float64)
        P, D = phi(x64), dgelu(x64)
        ph, dg = kernel_form(cp, xs, CL).astype(np.float64), kernel_form(cd, xs).astype(np.float64)
        eg, ep, ed = np.abs(x64 * ph - x64 * P), np.abs(ph - P), np.abs(dg - D)
        print(f"  {name}: |dGELU| {eg.max():.3e} (x = {x64[eg.argmax()]:+.3f})  |dPhi| {ep.max():.3e}  |dGELU'| {ed.max():.3e} (x = {x64[ed.argmax()]:+.3f})")
        if not name.startswith("tail"):
            worst = [max(worst[0], eg.max()), max(worst[1], ed.max())]
    print(f"  worst over the non-tail sets: GELU {worst[0]:.3e}, GELU' {worst[1]:.3e} -> {'PASS' if max(worst) <= 1e-3 else 'FAIL'} (bound 1e-3)")
    print("  (tail: GELU saturates at x * Phi_poly(+-4); the f32 polynomials of common.h have a 5e-4 GELU' tail of the same kind)")


if __name__ == "__main__":
    main()
    final()
