"""The reference's encrypted-CNN caller (package `cnn`, cnn/cnn.go) on the device-resident evaluator: the three layer
functions with the reference's names and argument order.  Every ciphertext, hoisted form and key stays in HBM between
the calls; the only host work is the float64 scale bookkeeping inside the mkckks mirror."""


def Convolution(eval, rlkSet, rtkSet, ctImage, ctImageHoisted, ctKernels, ctKernelsHoisted):
    """cnn.go:10-39: kernels pre-rotated by 0, 1, 14, 15; the image is hoisted once and reused by the three rotations"""
    convOut = eval.MulRelinHoistedNew(ctImage, ctKernels[0], ctImageHoisted, ctKernelsHoisted[0], rlkSet)
    for i, rot in ((1, 1), (2, 14), (3, 15)):
        temp = eval.RotateHoistedNew(ctImage, rot, ctImageHoisted, rtkSet)
        tempHoisted = eval.HoistedForm(temp)
        temp = eval.MulRelinHoistedNew(temp, ctKernels[i], tempHoisted, ctKernelsHoisted[i], rlkSet)
        convOut = eval.AddNew(convOut, temp)
    for rot in (2048, 1024):
        convOut = eval.AddNew(convOut, eval.RotateNew(convOut, rot, rtkSet))
    return convOut


def FC1Layer(eval, rlkSet, rtkSet, ctVec, ctVecHoisted, ctMat, ctMatHoisted, ctBias):
    """cnn.go:41-71: diagonal-packed 64 x 1024 matrix in 8 ciphertexts, then a log-sum over each 128-slot block"""
    fc1Out = None
    for i in range(len(ctMat)):
        temp = eval.RotateHoistedNew(ctVec, i * 128, ctVecHoisted, rtkSet)
        tempHoisted = eval.HoistedForm(temp)
        temp = eval.MulRelinHoistedNew(temp, ctMat[i], tempHoisted, ctMatHoisted[i], rlkSet)
        fc1Out = temp if i == 0 else eval.AddNew(fc1Out, temp)
    for i in range(7):                                        # log2(128)
        fc1Out = eval.AddNew(fc1Out, eval.RotateNew(fc1Out, 1 << i, rtkSet))
    return eval.AddNew(fc1Out, ctBias)


def FC2Layer(eval, rlkSet, rtkSet, ctVec, ctMat, ctBias, ptMask, ptMaskScale):
    """cnn.go:73-96; ptMask: the plaintext polynomial of the 0/1 mask (host, coefficient domain) and its scale"""
    fc2Out = eval.MulPtxtNew(ctVec, ptMask, ptMaskScale)
    for i in range(4):                                        # log2(16)
        fc2Out = eval.AddNew(fc2Out, eval.RotateNew(fc2Out, -(1 << i), rtkSet))
    fc2Out = eval.MulRelinNew(fc2Out, ctMat, rlkSet)
    for i in range(6):                                        # log2(64)
        fc2Out = eval.AddNew(fc2Out, eval.RotateNew(fc2Out, 128 * (1 << i), rtkSet))
    return eval.AddNew(fc2Out, ctBias)


def Inference(eval, rlkSet, rtkSet, ctImage, ctKernels, ctFC1, ctFC2, ctB1, ctB2, ptMask, ptMaskScale, hoisted=None):
    """the evaluation part of TestCNN / BenchmarkCNN (cnn_test.go:153-165): convolution, square, FC1, square, FC2.
    hoisted: optional (ctImageHoisted, ctKernelsHoisted, ctFC1Hoisted) precomputed by the caller, as the reference does."""
    if hoisted is None:
        hoisted = (eval.HoistedForm(ctImage), [eval.HoistedForm(c) for c in ctKernels], [eval.HoistedForm(c) for c in ctFC1])
    ctImageHoisted, ctKernelsHoisted, ctFC1Hoisted = hoisted
    convOut = Convolution(eval, rlkSet, rtkSet, ctImage, ctImageHoisted, ctKernels, ctKernelsHoisted)
    convOutHoisted = eval.HoistedForm(convOut)
    square1Out = eval.MulRelinHoistedNew(convOut, convOut, convOutHoisted, convOutHoisted, rlkSet)
    square1OutHoisted = eval.HoistedForm(square1Out)
    fc1Out = FC1Layer(eval, rlkSet, rtkSet, square1Out, square1OutHoisted, ctFC1, ctFC1Hoisted, ctB1)
    fc1OutHoisted = eval.HoistedForm(fc1Out)
    square2Out = eval.MulRelinHoistedNew(fc1Out, fc1Out, fc1OutHoisted, fc1OutHoisted, rlkSet)
    return FC2Layer(eval, rlkSet, rtkSet, square2Out, ctFC2, ctB2, ptMask, ptMaskScale)
