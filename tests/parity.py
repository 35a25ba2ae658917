"""Shared parity assertions for the model-level GPU tests: every comparison prints the MEASURED maximum delta (visible
in the GPU test log with -s / -rA) and asserts it at the north-star tolerance (1e-3), not at a blanket slack."""
import numpy as np

TOL = 1e-3            # BASELINE.json north_star: outputs within 1e-3 (fp32) of the reference CPU path


def maxdiff(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a - b).max()) if a.size else 0.0


def angle_diff(a, b):
    return np.abs((np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64) + 180.0) % 360.0 - 180.0)


def assert_text_prob_close(p, q, tol=TOL, tie_eps=1e-4, what="text", max_tied=0.05):
    """p (HIP) vs q (oracle) [R, T, C] character probabilities of a GREEDY decoder: step t+1 is fed step t's arg-max,
    so when the oracle's two best classes of a step are closer than `tie_eps` either choice is legitimate and the
    later steps of that RoI see a different input.  Such RoIs are compared up to and including that step only; their
    fraction is bounded.  Returns the measured max |dp|."""
    p, q = np.asarray(p, dtype=np.float64), np.asarray(q, dtype=np.float64)
    assert p.shape == q.shape, (p.shape, q.shape)
    R, T, _ = q.shape
    if R == 0:
        return 0.0
    srt = np.sort(q, axis=-1)
    live = q.sum(-1) > 0                                           # steps before the early break
    near = ((srt[..., -1] - srt[..., -2]) < tie_eps) & live
    first = np.where(near.any(1), near.argmax(1), T)
    mask = np.arange(T)[None, :] <= first[:, None]
    err = np.abs(p - q)
    worst = float(err[mask].max())
    agree = float((p.argmax(-1)[mask] == q.argmax(-1)[mask]).mean())
    tied = float((first < T).mean())
    print(f"[parity] {what}: max |dp| = {worst:.3e} over {int(mask.sum())} (RoI, step) pairs, arg-max agreement {agree:.4f}, "
          f"RoIs cut at a near-tie (< {tie_eps:g}): {int((first < T).sum())}/{R}")
    assert worst < tol, f"{what}: max |dp| {worst:.3e} >= {tol}"
    assert tied <= max_tied, f"{what}: {tied:.3f} of the RoIs have a near-tie"
    # away from ties the decoded characters are identical
    far = ~near & mask & live
    assert (p.argmax(-1)[far] == q.argmax(-1)[far]).all()
    return worst


def assert_detections_close(got, ref, tol=TOL, what="detections", box_atol=2e-3, box_rtol=1e-4):
    """got: dict(scores, boxes, orientations|None, kept|None) from the HIP path, ref: the oracle's dict
    (scores, pred_boxes, orientations, kept).  Same count, same kept proposal indices in the same order, scores and
    orientation probabilities within `tol`, boxes within box_atol + box_rtol*|x| pixels (angles modulo 360)."""
    n_ref = len(ref["scores"])
    assert len(got["scores"]) == n_ref, f"{what}: {len(got['scores'])} detections vs oracle {n_ref}"
    if n_ref == 0:
        print(f"[parity] {what}: 0 detections on both sides")
        return
    if got.get("kept") is not None and ref.get("kept") is not None:
        gk, rk = np.asarray(got["kept"]).tolist(), np.asarray(ref["kept"]).tolist()
        assert gk == rk, f"{what}: kept proposal indices differ: {gk} vs {rk}"
    ds = maxdiff(got["scores"], ref["scores"])
    gb, rb = np.asarray(got["boxes"], dtype=np.float64), np.asarray(ref["pred_boxes"], dtype=np.float64)
    db = np.abs(gb - rb)
    db[:, 4] = angle_diff(gb[:, 4], rb[:, 4])
    msg = f"[parity] {what}: n = {n_ref}, max |dscore| = {ds:.3e}, max |dbox| = {db.max():.3e} px/deg"
    assert ds < tol, f"{what}: max |dscore| {ds:.3e}"
    assert (db <= box_atol + box_rtol * np.abs(rb)).all(), f"{what}: max |dbox| {db.max():.3e}"
    if got.get("orientations") is not None and ref.get("orientations") is not None:
        go, ro = np.asarray(got["orientations"]), np.asarray(ref["orientations"])
        assert (go[:, 0] == ro[:, 0]).all(), f"{what}: orientation classes differ"
        do = maxdiff(go[:, 1], ro[:, 1])
        msg += f", max |dorientation prob| = {do:.3e}"
        assert do < tol
    print(msg)


def assert_same_box_set(got, ref, atol=2e-3, rtol=1e-4):
    """order-insensitive match (near-tied scores may swap neighbours); angles compared modulo 360."""
    got, ref = np.asarray(got, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    assert got.shape == ref.shape, (got.shape, ref.shape)
    used = set()
    for i, r in enumerate(ref):
        d = np.abs(got - r)
        d[:, 4] = np.abs((got[:, 4] - r[4] + 180.0) % 360.0 - 180.0)
        ok = (d <= atol + rtol * np.abs(r)).all(axis=1)
        cand = [j for j in np.nonzero(ok)[0] if j not in used]
        assert cand, f"oracle box {i} {r} has no HIP match (closest {got[d.sum(1).argmin()]})"
        j = min(cand, key=lambda j: abs(j - i))
        assert abs(j - i) <= 3, f"box {i} matched far away at {j}"
        used.add(j)


def match_box_sets(got_boxes, got_scores, ref_boxes, ref_scores, atol_px=0.5, rtol=1e-2):
    """order-free greedy matching of two rotated-box sets (reduced-precision runs reorder near-tied scores and may
    gain / lose a box at a threshold).  Returns (fraction of ref matched, fraction of got matched, max |dscore| over the
    matches, max |dbox| over the matches)."""
    gb, rb = np.asarray(got_boxes, dtype=np.float64), np.asarray(ref_boxes, dtype=np.float64)
    gs, rs = np.asarray(got_scores, dtype=np.float64), np.asarray(ref_scores, dtype=np.float64)
    if len(rb) == 0 or len(gb) == 0:
        return (1.0 if len(rb) == 0 else 0.0), (1.0 if len(gb) == 0 else 0.0), 0.0, 0.0
    used = np.zeros(len(gb), dtype=bool)
    ds, db, hit = 0.0, 0.0, 0
    for i, r in enumerate(rb):
        d = np.abs(gb - r)
        d[:, 4] = angle_diff(gb[:, 4], r[4])
        ok = (d <= atol_px + rtol * np.abs(r)).all(axis=1) & ~used
        if not ok.any():
            continue
        j = int(np.argmin(np.where(ok, d.sum(1), np.inf)))
        used[j] = True
        hit += 1
        ds = max(ds, abs(gs[j] - rs[i]))
        db = max(db, float(d[j].max()))
    return hit / len(rb), float(used.mean()), ds, db


def teacher_forced_text_probs(dec, enc, q):
    """The product's decoder, one step at a time (glass_attention_decode_step: additive attention, GRU cell, fc, soft-max - the
    arithmetic of the one-launch decoder), FED THE ORACLE'S previous symbols: step i of RoI r sees arg-max q[r, i-1] instead of its
    own arg-max.  A greedy decoder turns an ulp-sized difference at a near-tie into a different word from there on; teacher-forced,
    both sides walk the same symbol sequence and EVERY live step of EVERY RoI is comparable (no `max_tied`).
    dec: ASTER_V2 with weights; enc [R,T,D] device tensor (the product's own encoder output); q [R,L,C] oracle probabilities
    (rows of steps after the oracle's early break are zero and stay zero here).  Returns [R,L,C] numpy."""
    import torch
    from glass_amd.ops import native as K
    q = np.asarray(q)
    R, L, C = q.shape
    out = np.zeros_like(q, dtype=np.float32)
    if R == 0:
        return out
    enc = enc.contiguous()
    T, D = enc.shape[1], enc.shape[2]
    xproj = K.linear(enc.view(R * T, D), dec.w["xW"], dec.w["xB"]).view(R, T, D)
    h = torch.zeros((R, D), dtype=torch.float32, device=enc.device)
    y_prev = torch.zeros((R,), dtype=torch.int32, device=enc.device)
    for i in range(L):
        live = q[:, i].sum(-1) > 0
        if not live.any():
            break
        _, probs, h = K.attention_decode_step(enc, xproj, dec.w, h, y_prev, C)
        out[live, i] = probs.cpu().numpy()[live]
        y_prev = torch.from_numpy(q[:, i].argmax(-1).astype(np.int32)).to(enc.device)
    return out


def text_prob_stats(p, q, what="text"):
    """p vs q [R,L,C] over the live steps of q: (max |dp|, mean |dp|, 95th percentile of the per-step max |dp|, arg-max agreement)"""
    p, q = np.asarray(p, dtype=np.float64), np.asarray(q, dtype=np.float64)
    live = q.sum(-1) > 0
    d = np.abs(p - q)[live]
    step_max = d.max(-1)
    agree = float((p.argmax(-1)[live] == q.argmax(-1)[live]).mean())
    st = (float(d.max()), float(d.mean()), float(np.percentile(step_max, 95)), agree)
    print(f"[parity] {what}: {int(live.sum())} live (RoI, step) pairs: max |dp| {st[0]:.3e}, mean |dp| {st[1]:.3e}, p95 of per-step max |dp| {st[2]:.3e}, "
          f"arg-max agreement {st[3]:.4f}")
    return st
