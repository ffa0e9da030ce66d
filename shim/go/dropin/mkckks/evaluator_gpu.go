// evaluator_gpu.go -- mkckks.GPUEvaluator: the reference's Evaluator with its key-switching methods as SINGLE engine calls.
// NOT BUILT OR TESTED IN THIS REPOSITORY (no Go toolchain); tests/test_go_dropin_static.py holds every method below against the signature of
// the mkckks.Evaluator method of the same name (tests/golden/ref_go_signatures.json).
//
// With shim/go/dropin/mkrlwe/keyswitch_gpu.go installed, the reference's own mkckks.Evaluator already runs on the GPU (it reaches the polynomial
// layer only through mkrlwe.KeySwitcher) -- its tests and cnn/ need no change.  What that leaves on the table is structure: MulRelinNew there is
// 2k Decompose calls (one upload each), one MulAndRelinHoisted and a Rescale on the host (evaluator.go:416-443,570-581,359-398).  GPUEvaluator
// embeds *Evaluator -- so AddNew, SubNew, MultByConst, MulPtxtNew, DropLevel[New], Rescale[New] are the reference's own code on host ciphertexts,
// with their signatures -- and redefines the methods below, each as: upload the operands, ONE engine call (hoisting, MulAndRelin and the
// Rescale fused: mkhe_mul_relin_rescale), download the result.  The benchmark's timed region (mkckks_benchmark_test.go:78-82) is then the
// engine's 0.75 ms plus two uploads and one download.  To use it in the reference's tests apply shim/go/patches/mkckks_tests_gpu_evaluator.diff
// (the evaluator field of testParams and its constructor: two lines).
//
//go:build mkhe_gpu
// +build mkhe_gpu

package mkckks

import (
	"mk-lattigo/mkrlwe"
	"mk-lattigo/mkrlwegpu"
)

// GPUEvaluator: see the file comment.  Not reentrant, like Evaluator (shared pools, one engine stream).
type GPUEvaluator struct {
	*Evaluator
}

// NewGPUEvaluator creates the evaluator (NewEvaluator, evaluator.go:23-38) and switches the host mirror of hoisted forms off: this type only
// passes them on (HoistedForm -> MulRelinHoistedNew / RotateHoistedNew); eval.KeySwitcher().Materialize(swk) brings one to the host.
func NewGPUEvaluator(params Parameters) *GPUEvaluator {
	eval := &GPUEvaluator{NewEvaluator(params)}
	eval.ksw.HostMirror = false
	return eval
}

// KeySwitcher exposes the mkrlwe.KeySwitcher (residency cache: Forget, Materialize, Resident).
func (eval *GPUEvaluator) KeySwitcher() *mkrlwe.KeySwitcher { return eval.ksw }

func (eval *GPUEvaluator) gpu() *mkrlwegpu.Context { return eval.ksw.GPU() }

// rescaleCount is the loop of Rescale (evaluator.go:376-384): how many moduli the scale is divided by, and the scale after it.
func (eval *GPUEvaluator) rescaleCount(level int, scale, minScale float64) (nb int, out float64) {
	q := eval.params.RingQ().Modulus
	out = scale
	for level-nb >= 0 && out/float64(q[level-nb]) >= minScale/2 { // (the bound first: the reference indexes before it checks)
		out /= float64(q[level-nb])
		nb++
	}
	return
}

func minInt(a, b int) int {
	if a < b {
		return a
	}
	return b
}

func hoistedKeys(ks *mkrlwe.KeySwitcher, h *mkrlwe.HoistedCiphertext, ids []string) []*mkrlwegpu.SwitchingKey {
	if h == nil {
		return nil
	}
	out := make([]*mkrlwegpu.SwitchingKey, len(ids))
	for i, id := range ids {
		swk, ok := h.Value[id]
		if !ok {
			panic("mkckks (gpu): the hoisted ciphertext lacks id " + id)
		}
		out[i] = ks.Resident(swk)
	}
	return out
}

func (eval *GPUEvaluator) relinKeys(rlkSet *mkrlwe.RelinearizationKeySet, idLists ...[]string) mkrlwegpu.RelinKeys {
	rk := mkrlwegpu.RelinKeys{}
	for _, ids := range idLists {
		for _, id := range ids {
			if _, ok := rk[id]; !ok {
				rlk := rlkSet.GetRelinearizationKey(id)
				rk[id] = [3]*mkrlwegpu.SwitchingKey{eval.ksw.Resident(rlk.Value[0]), eval.ksw.Resident(rlk.Value[1]), eval.ksw.Resident(rlk.Value[2])}
			}
		}
	}
	return rk
}

// MulRelinNew (evaluator.go:416-443): the engine hoists both operands itself.
func (eval *GPUEvaluator) MulRelinNew(op0, op1 *Ciphertext, rlkSet *mkrlwe.RelinearizationKeySet) (ctOut *Ciphertext) {
	return eval.MulRelinHoistedNew(op0, op1, nil, nil, rlkSet)
}

// MulRelinHoistedNew (evaluator.go:558-581): MulAndRelinHoisted and the Rescale that always follows it, in one engine call when the scale
// asks for the usual single division (mkhe_mul_relin_rescale); otherwise the product at its level and nb divisions on the device.
func (eval *GPUEvaluator) MulRelinHoistedNew(op0, op1 *Ciphertext, op0Hoisted, op1Hoisted *mkrlwe.HoistedCiphertext, rlkSet *mkrlwe.RelinearizationKeySet) (ctOut *Ciphertext) {
	g := eval.gpu()
	level := minInt(op0.Level(), op1.Level())
	scale := op0.ScalingFactor() * op1.ScalingFactor()
	idset := op0.IDSet().Union(op1.IDSet())
	d0 := g.Upload(op0.Value, op0.Level())
	defer d0.Close()
	d1 := d0
	if op1 != op0 {
		d1 = g.Upload(op1.Value, op1.Level())
		defer d1.Close()
	}
	h0, h1 := hoistedKeys(eval.ksw, op0Hoisted, d0.IDs()), hoistedKeys(eval.ksw, op1Hoisted, d1.IDs())
	rk := eval.relinKeys(rlkSet, d0.IDs(), d1.IDs())
	crsU := eval.ksw.Resident(eval.params.CRS[-1])
	nb, outScale := 0, scale
	if level > 0 && scale != 0 && eval.params.Scale() > 0 { // the conditions under which Rescale acts at all (evaluator.go:363-373)
		nb, outScale = eval.rescaleCount(level, scale, eval.params.Scale())
	}
	ctOut = NewCiphertext(eval.params, idset, level-nb, outScale)
	ids := mkrlwegpu.SortedIDs(ctOut.Value)
	if nb == 1 {
		out := g.NewCiphertext(ids, level-1)
		defer out.Close()
		g.MulRelinRescale(d0, d1, h0, h1, rk, crsU, out)
		g.Download(out, ctOut.Value, level-1)
		return
	}
	prod := g.NewCiphertext(ids, level)
	defer prod.Close()
	g.MulAndRelinHoisted(d0, d1, h0, h1, rk, crsU, prod)
	if nb == 0 {
		g.Download(prod, ctOut.Value, level)
		return
	}
	out := g.NewCiphertext(ids, level-nb)
	defer out.Close()
	g.Rescale(prod, nb, out)
	g.Download(out, ctOut.Value, level-nb)
	return
}

// HoistedForm (evaluator.go:543-553): one batched launch for all party components; the digit vectors stay on the device, bound to host
// SwitchingKeys whose Value is empty (KeySwitcher().Materialize fills one in).
func (eval *GPUEvaluator) HoistedForm(ct *Ciphertext) (ctHoisted *mkrlwe.HoistedCiphertext) {
	g := eval.gpu()
	d := g.Upload(ct.Value, ct.Level())
	defer d.Close()
	ctHoisted = mkrlwe.NewHoistedCiphertext()
	keys := g.HoistedForm(d, ct.Level())
	for i, id := range d.IDs() {
		swk := new(mkrlwe.SwitchingKey)
		eval.ksw.Adopt(swk, keys[i])
		ctHoisted.Value[id] = swk
	}
	return
}

func (eval *GPUEvaluator) normRot(rotidx int) int {
	n2 := eval.params.N() / 2
	for rotidx >= n2 {
		rotidx -= n2
	}
	for rotidx < 0 {
		rotidx += n2
	}
	return rotidx
}

func (eval *GPUEvaluator) rotationKeys(rkSet *mkrlwe.RotationKeySet, ids []string, rotidx int) []*mkrlwegpu.SwitchingKey {
	out := make([]*mkrlwegpu.SwitchingKey, len(ids))
	for i, id := range ids {
		out[i] = eval.ksw.Resident(rkSet.GetRotationKey(id, uint(rotidx)).Value)
	}
	return out
}

// RotateNew (evaluator.go:485-525): one engine call for an index with a CRS; otherwise the reference's walk over the powers of two, the
// intermediate ciphertexts staying on the device.
func (eval *GPUEvaluator) RotateNew(ct0 *Ciphertext, rotidx int, rkSet *mkrlwe.RotationKeySet) (ctOut *Ciphertext) {
	ctOut = NewCiphertext(eval.params, ct0.IDSet(), ct0.Level(), ct0.Scale)
	rotidx = eval.normRot(rotidx)
	if rotidx == 0 {
		ctOut.Ciphertext.Copy(ct0.Ciphertext)
		return
	}
	g := eval.gpu()
	cur := g.Upload(ct0.Value, ct0.Level())
	steps := []int{rotidx}
	if _, in := eval.params.CRS[rotidx]; !in {
		steps = steps[:0]
		for k := 1; rotidx > 0; k *= 2 {
			if rotidx%2 != 0 {
				steps = append(steps, k)
			}
			rotidx /= 2
		}
	}
	for _, r := range steps {
		crs, ok := eval.params.CRS[r]
		if !ok {
			panic("mkckks (gpu): no CRS for rotation index")
		}
		next := g.NewCiphertext(cur.IDs(), ct0.Level())
		g.RotateHoisted(cur, r, nil, eval.rotationKeys(rkSet, cur.IDs(), r), eval.ksw.Resident(crs), next)
		cur.Close()
		cur = next
	}
	g.Download(cur, ctOut.Value, ct0.Level())
	cur.Close()
	return
}

// RotateHoistedNew (evaluator.go:585-617).
func (eval *GPUEvaluator) RotateHoistedNew(ct0 *Ciphertext, rotidx int, ct0Hoisted *mkrlwe.HoistedCiphertext, rkSet *mkrlwe.RotationKeySet) (ctOut *Ciphertext) {
	ctOut = NewCiphertext(eval.params, ct0.IDSet(), ct0.Level(), ct0.Scale)
	rotidx = eval.normRot(rotidx)
	if rotidx == 0 {
		ctOut.Ciphertext.Copy(ct0.Ciphertext)
		return
	}
	crs, in := eval.params.CRS[rotidx]
	if !in {
		panic("Hoisted rotation only works for precomputed rotation keys")
	}
	g := eval.gpu()
	d := g.Upload(ct0.Value, ct0.Level())
	defer d.Close()
	out := g.NewCiphertext(d.IDs(), ct0.Level())
	defer out.Close()
	g.RotateHoisted(d, rotidx, hoistedKeys(eval.ksw, ct0Hoisted, d.IDs()), eval.rotationKeys(rkSet, d.IDs(), rotidx), eval.ksw.Resident(crs), out)
	g.Download(out, ctOut.Value, ct0.Level())
	return
}

// ConjugateNew (evaluator.go:530-541).
func (eval *GPUEvaluator) ConjugateNew(ct0 *Ciphertext, ckSet *mkrlwe.ConjugationKeySet) (ctOut *Ciphertext) {
	ctOut = NewCiphertext(eval.params, ct0.IDSet(), ct0.Level(), ct0.Scale)
	eval.ksw.Conjugate(ct0.Ciphertext, ckSet, ctOut.Ciphertext)
	return
}

// RescaleNew (evaluator.go:406-412, Rescale :359-398): the divisions by the last moduli on the device.
func (eval *GPUEvaluator) RescaleNew(ct0 *Ciphertext, threshold float64) (ctOut *Ciphertext, err error) {
	if threshold <= 0 || ct0.Scale == 0 || ct0.Level() == 0 {
		return eval.Evaluator.RescaleNew(ct0, threshold) // the error cases, reported by the reference's own code
	}
	nb, scale := eval.rescaleCount(ct0.Level(), ct0.Scale, threshold)
	if nb == 0 {
		return eval.Evaluator.RescaleNew(ct0, threshold)
	}
	g := eval.gpu()
	d := g.Upload(ct0.Value, ct0.Level())
	defer d.Close()
	out := g.NewCiphertext(d.IDs(), ct0.Level()-nb)
	defer out.Close()
	g.Rescale(d, nb, out)
	ctOut = NewCiphertext(eval.params, ct0.IDSet(), ct0.Level()-nb, scale)
	g.Download(out, ctOut.Value, ct0.Level()-nb)
	return ctOut, nil
}
