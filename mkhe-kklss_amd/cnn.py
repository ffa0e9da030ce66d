"""The reference's encrypted-CNN caller (package `cnn`, cnn/cnn.go) on the device-resident evaluator: the three layer
functions with the reference's names and argument order.  Every ciphertext, hoisted form and key stays in HBM between
the calls; the only host work is the float64 scale bookkeeping inside the mkckks mirror.

`forks` (optional, a list of mkckks.Evaluator.Fork() evaluators): the independent rotate -> hoist -> MulRelin chains of
Convolution (3) and FC1Layer (8) are issued through different engine contexts and overlap on the GPU -- every launch of
this workload covers a few dozen 2^14-point limbs, a fraction of the 256 CUs.  Results are the same bit for bit.

Round 5: an evaluator that offers `Lanes(n)` (the device mkckks.Evaluator) runs those independent chains as the LANES of one launch set instead
(BatchEvaluator over the same context: one rotation launch for the 3 resp. 7 rotated copies, each with its own Galois element and keys, one hoisting
launch, one MulRelin launch set for the 4 resp. 8 products) -- a kernel trace showed that the small kernels of forked chains do not overlap on this
chip, so 8 chains cost 8 times the launches -- and one that offers `RotateAndAddNew` folds the AddNew of every "rotate, then add" step of the log-sums
into the rotation's store.  Evaluators without them (the CPU evaluator of tests/, a BatchEvaluator over B images) take the reference's sequence
of calls unchanged; the device result equals it bit for bit (tests/test_gpu_cnn.py)."""


def _fan_out(eval, used):
    for f in used:
        f.params.wait_for(eval.params)


def _fan_in(eval, used):
    for f in used:
        eval.params.wait_for(f.params)


def _rot_add(eval, ct, rot, rtkSet):
    """temp = eval.RotateNew(ct, rot, rtkSet); ct = eval.AddNew(ct, temp)   (cnn.go:33-37,64-67,83-86,90-93)"""
    fused = getattr(eval, "RotateAndAddNew", None)
    return fused(ct, rot, rtkSet) if fused else eval.AddNew(ct, eval.RotateNew(ct, rot, rtkSet))


def _sum(eval, cts):
    """out = cts[0]; for c in cts[1:]: out = eval.AddNew(out, c)   (cnn.go:19-30,58-62)"""
    fused = getattr(eval, "SumNew", None)
    if fused:
        return fused(cts)
    out = cts[0]
    for c in cts[1:]:
        out = eval.AddNew(out, c)
    return out


def _lane_products(eval, rlkSet, rtkSet, ct, ctHoisted, rots, ctOther, ctOtherHoisted):
    """[MulRelinHoistedNew(Rot_r(ct), ctOther[i], HoistedForm(Rot_r(ct)), ctOtherHoisted[i]) for i, r in enumerate(rots)] as lanes of one launch set;
    r = 0: the rotation is a copy and its hoisted form IS ctHoisted (cnn.go:16, :53 with i = 0)"""
    from .mkckks import BatchCiphertext, BatchHoisted
    moving = [r for r in rots if r != 0]
    if isinstance(ct, BatchCiphertext):
        # B images in lock step: lane-major item lists of B * n (item j * B + b = lane j of image b); the model operands are the same for every image
        B = len(ct.cts)
        temps, tempsH = [], []
        if moving:
            lanes = eval.Lanes(len(moving))
            t = lanes.RotateHoistedNew(BatchCiphertext([c for _ in moving for c in ct.cts]), [r for r in moving for _ in range(B)],
                                       BatchHoisted([h for _ in moving for h in ctHoisted.hoisted]), rtkSet)
            temps, tempsH = t.cts, lanes.HoistedForm(t).hoisted
        ops, hs, k = [], [], 0
        for r in rots:
            if r == 0:
                ops += ct.cts; hs += ctHoisted.hoisted
            else:
                ops += temps[k * B:(k + 1) * B]; hs += tempsH[k * B:(k + 1) * B]; k += 1
        prods = eval.Lanes(len(rots)).MulRelinHoistedNew(BatchCiphertext(ops), BatchCiphertext([c for c in ctOther for _ in range(B)]), BatchHoisted(hs),
                                                         BatchHoisted([h for h in ctOtherHoisted for _ in range(B)]), rlkSet)
        return [BatchCiphertext(prods.cts[i * B:(i + 1) * B]) for i in range(len(rots))]
    temps, tempsH = [], []
    if moving:
        lanes = eval.Lanes(len(moving))
        t = lanes.RotateHoistedNew(ct, moving, ctHoisted, rtkSet)
        temps, tempsH = t.cts, lanes.HoistedForm(t).hoisted
    ops, hs, k = [], [], 0
    for r in rots:
        if r == 0:
            ops.append(ct); hs.append(ctHoisted)
        else:
            ops.append(temps[k]); hs.append(tempsH[k]); k += 1
    prods = eval.Lanes(len(rots)).MulRelinHoistedNew(BatchCiphertext(ops), BatchCiphertext(list(ctOther)), BatchHoisted(hs), BatchHoisted(list(ctOtherHoisted)), rlkSet)
    return prods.cts


def Convolution(eval, rlkSet, rtkSet, ctImage, ctImageHoisted, ctKernels, ctKernelsHoisted, forks=None):
    """cnn.go:10-39: kernels pre-rotated by 0, 1, 14, 15; the image is hoisted once and reused by the three rotations"""
    if not forks and hasattr(eval, "Lanes"):
        prods = _lane_products(eval, rlkSet, rtkSet, ctImage, ctImageHoisted, (0, 1, 14, 15), ctKernels, ctKernelsHoisted)
        convOut = _sum(eval, prods)
        for rot in (2048, 1024):
            convOut = _rot_add(eval, convOut, rot, rtkSet)
        return convOut
    def chain(ev, i, rot):
        temp = ev.RotateHoistedNew(ctImage, rot, ctImageHoisted, rtkSet)
        tempHoisted = ev.HoistedForm(temp)
        return ev.MulRelinHoistedNew(temp, ctKernels[i], tempHoisted, ctKernelsHoisted[i], rlkSet)
    work = ((1, 1), (2, 14), (3, 15))
    used = [forks[j % len(forks)] for j in range(len(work))] if forks else []
    _fan_out(eval, set(used))
    convOut = eval.MulRelinHoistedNew(ctImage, ctKernels[0], ctImageHoisted, ctKernelsHoisted[0], rlkSet)
    temps = [chain(used[j] if forks else eval, i, rot) for j, (i, rot) in enumerate(work)]
    _fan_in(eval, set(used))
    for temp in temps:
        convOut = eval.AddNew(convOut, temp)
    for rot in (2048, 1024):
        convOut = _rot_add(eval, convOut, rot, rtkSet)
    return convOut


def FC1Layer(eval, rlkSet, rtkSet, ctVec, ctVecHoisted, ctMat, ctMatHoisted, ctBias, forks=None):
    """cnn.go:41-71: diagonal-packed 64 x 1024 matrix in 8 ciphertexts, then a log-sum over each 128-slot block"""
    def chain(ev, i):
        temp = ev.RotateHoistedNew(ctVec, i * 128, ctVecHoisted, rtkSet)
        tempHoisted = ev.HoistedForm(temp)
        return ev.MulRelinHoistedNew(temp, ctMat[i], tempHoisted, ctMatHoisted[i], rlkSet)
    if not forks and hasattr(eval, "Lanes"):
        temps = _lane_products(eval, rlkSet, rtkSet, ctVec, ctVecHoisted, [i * 128 for i in range(len(ctMat))], ctMat, ctMatHoisted)
    else:
        evs = [eval] + list(forks or [])
        used = set(evs[i % len(evs)] for i in range(len(ctMat))) - {eval}
        _fan_out(eval, used)
        temps = [chain(evs[i % len(evs)], i) for i in range(len(ctMat))]
        _fan_in(eval, used)
    fc1Out = _sum(eval, temps)
    for i in range(7):                                        # log2(128)
        fc1Out = _rot_add(eval, fc1Out, 1 << i, rtkSet)
    return eval.AddNew(fc1Out, ctBias)


def FC2Layer(eval, rlkSet, rtkSet, ctVec, ctMat, ctBias, ptMask, ptMaskScale):
    """cnn.go:73-96; ptMask: the plaintext polynomial of the 0/1 mask (host, coefficient domain) and its scale"""
    fc2Out = eval.MulPtxtNew(ctVec, ptMask, ptMaskScale)
    for i in range(4):                                        # log2(16)
        fc2Out = _rot_add(eval, fc2Out, -(1 << i), rtkSet)
    fc2Out = eval.MulRelinNew(fc2Out, ctMat, rlkSet)
    for i in range(6):                                        # log2(64)
        fc2Out = _rot_add(eval, fc2Out, 128 * (1 << i), rtkSet)
    return eval.AddNew(fc2Out, ctBias)


def Inference(eval, rlkSet, rtkSet, ctImage, ctKernels, ctFC1, ctFC2, ctB1, ctB2, ptMask, ptMaskScale, hoisted=None, forks=None):
    """the evaluation part of TestCNN / BenchmarkCNN (cnn_test.go:153-165): convolution, square, FC1, square, FC2.
    hoisted: optional (ctImageHoisted, ctKernelsHoisted, ctFC1Hoisted) precomputed by the caller, as the reference does."""
    if hoisted is None:
        hoisted = (eval.HoistedForm(ctImage), [eval.HoistedForm(c) for c in ctKernels], [eval.HoistedForm(c) for c in ctFC1])
    ctImageHoisted, ctKernelsHoisted, ctFC1Hoisted = hoisted
    convOut = Convolution(eval, rlkSet, rtkSet, ctImage, ctImageHoisted, ctKernels, ctKernelsHoisted, forks)
    convOutHoisted = eval.HoistedForm(convOut)
    square1Out = eval.MulRelinHoistedNew(convOut, convOut, convOutHoisted, convOutHoisted, rlkSet)
    square1OutHoisted = eval.HoistedForm(square1Out)
    fc1Out = FC1Layer(eval, rlkSet, rtkSet, square1Out, square1OutHoisted, ctFC1, ctFC1Hoisted, ctB1, forks)
    fc1OutHoisted = eval.HoistedForm(fc1Out)
    square2Out = eval.MulRelinHoistedNew(fc1Out, fc1Out, fc1OutHoisted, fc1OutHoisted, rlkSet)
    return FC2Layer(eval, rlkSet, rtkSet, square2Out, ctFC2, ctB2, ptMask, ptMaskScale)
