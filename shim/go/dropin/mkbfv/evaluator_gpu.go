// evaluator_gpu.go -- mkbfv.GPUEvaluator: the reference's BFV Evaluator with MulRelinNew as ONE engine call (mkhe_bfv_mul_relin: ModUpQtoR and
// Rescale of the operands, DecomposeBFV, tensor over R, Quantize and the double-gadget relinearization; mkbfv/evaluator.go:84-150,
// keyswitch_hoisted.go:39-207, basis_extension.go:49-97).  NOT BUILT OR TESTED IN THIS REPOSITORY (no Go toolchain); signatures held against
// the reference's by tests/test_go_dropin_static.py.
//
// With shim/go/dropin/mkrlwe/keyswitch_gpu.go installed the reference's mkbfv.Evaluator compiles and runs unchanged: its KeySwitcher embeds
// *mkrlwe.KeySwitcher (mkbfv/keyswitch.go:6-9), so Rotate / Conjugate / Decompose / ExternalProduct[Hoisted] run on the engine and the
// R-basis work stays host code.  GPUEvaluator embeds *Evaluator -- AddNew, SubNew, RotateNew, ConjugateNew are the reference's methods with their
// signatures (RotateNew / ConjugateNew reach the engine through the key switcher) -- and redefines MulRelinNew on a BFV engine context of its own
// (rings Q, QMul, P and the plaintext modulus).  In the reference's tests: shim/go/patches/mkbfv_tests_gpu_evaluator.diff (two lines).
//
//go:build mkhe_gpu
// +build mkhe_gpu

package mkbfv

import (
	"os"
	"strconv"

	"mk-lattigo/mkrlwe"
	"mk-lattigo/mkrlwegpu"
)

type GPUEvaluator struct {
	*Evaluator
	bfv  *mkrlwegpu.Context
	keys *mkrlwegpu.BFVRelinKeys
	have map[*RelinearizationKey]bool // relinearization keys uploaded so far, by pointer (immutable once used; ForgetKeys drops them all)
	crsU *mkrlwegpu.SwitchingKey
}

func NewGPUEvaluator(params Parameters) *GPUEvaluator {
	device := 0
	if v, err := strconv.Atoi(os.Getenv("MKHE_GO_DEVICE")); err == nil {
		device = v
	}
	eval := &GPUEvaluator{Evaluator: NewEvaluator(params)}
	eval.bfv = mkrlwegpu.NewBFVContext(params, device)
	eval.keys = mkrlwegpu.NewBFVRelinKeys()
	eval.have = map[*RelinearizationKey]bool{}
	return eval
}

// ForgetKeys drops every uploaded relinearization key (keys that were regenerated in place).
func (eval *GPUEvaluator) ForgetKeys() {
	for _, m := range []map[string]*mkrlwegpu.SwitchingKey{eval.keys.B1, eval.keys.B2, eval.keys.D1, eval.keys.D2, eval.keys.V} {
		for id, k := range m {
			k.Close()
			delete(m, id)
		}
	}
	eval.have = map[*RelinearizationKey]bool{}
}

func (eval *GPUEvaluator) ensureKeys(rlkSet *RelinearizationKeySet, ids []string) {
	for _, id := range ids {
		rlk := rlkSet.GetRelinearizationKey(id) // panics like the reference when the key is missing (keys.go:75-83)
		if eval.have[rlk] {
			continue
		}
		// rlk.Value[g].Value[k]: gadget g (Q digits, QMul digits), k = 0 b, 1 d, 2 v (keys.go:6-9, keygen.go:24-88)
		eval.bfv.UploadBFVRelinKey(eval.keys, id, rlk.Value[0].Value[0].Value, rlk.Value[1].Value[0].Value,
			rlk.Value[0].Value[1].Value, rlk.Value[1].Value[1].Value, rlk.Value[0].Value[2].Value)
		eval.have[rlk] = true
	}
}

// MulRelinNew (mkbfv/evaluator.go:84-88 -> mulRelinHoisted :118-150).
func (eval *GPUEvaluator) MulRelinNew(op0, op1 *Ciphertext, rlkSet *RelinearizationKeySet) (ctOut *Ciphertext) {
	ctOut = NewCiphertext(eval.params, op0.IDSet().Union(op1.IDSet()))
	g := eval.bfv
	level := eval.params.MaxLevel()
	d0 := g.Upload(op0.Value, level)
	defer d0.Close()
	d1 := d0
	if op1 != op0 {
		d1 = g.Upload(op1.Value, level)
		defer d1.Close()
	}
	eval.ensureKeys(rlkSet, d0.IDs())
	eval.ensureKeys(rlkSet, d1.IDs())
	if eval.crsU == nil {
		eval.crsU = g.UploadSwitchingKey(eval.params.CRS[-1].Value)
	}
	out := g.NewCiphertext(mkrlwegpu.SortedIDs(ctOut.Value), level)
	defer out.Close()
	g.MulRelinBFV(d0, d1, d0.IDs(), d1.IDs(), eval.keys, eval.crsU, out)
	g.Download(out, ctOut.Value, level)
	return
}

var _ = mkrlwe.NewIDSet
