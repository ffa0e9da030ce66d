// mkbfvgpu.go -- cgo binding of the mkhe_bfv_* entry points (include/mkhe.h) for the reference package mkbfv.
//
// NOT BUILT OR TESTED IN THIS REPOSITORY (no Go toolchain; see mkrlwegpu.go).  It shows the binding a maintainer
// adds so that mkbfv.Evaluator.MulRelinNew (mkbfv/evaluator.go:78-82) runs on an MI355X.
//
//go:build mkhe_gpu
// +build mkhe_gpu

package mkrlwegpu

/*
#include <stdlib.h>
#include "mkhe.h"
*/
import "C"

import (
	"runtime"
	"unsafe"

	"github.com/ldsec/lattigo/v2/ring"
	"github.com/ldsec/lattigo/v2/rlwe"
)

// BFVParams is what the binding reads of mkbfv.Parameters (mkbfv/params.go:27-34,83-103) beyond Params.
type BFVParams interface {
	Params
	RingQMul() *ring.Ring
	T() uint64
}

// NewBFVContext replaces mkbfv.NewKeySwitcher + NewFastBasisExtender (mkbfv/keyswitch.go:31-65,
// basis_extension.go:20-47).  The engine generates lattigo's default roots itself (ring.NewRing rule).
func NewBFVContext(params BFVParams, device int) *Context {
	q, qm, p := params.RingQ().Modulus, params.RingQMul().Modulus, params.RingP().Modulus
	ctx := &Context{params: params, ids: map[string]C.int{}}
	must(C.mkhe_ctx_create_bfv(&ctx.c, C.int(params.LogN()),
		(*C.uint64_t)(unsafe.Pointer(&q[0])), (*C.uint64_t)(unsafe.Pointer(&qm[0])), C.int(len(q)),
		(*C.uint64_t)(unsafe.Pointer(&p[0])), C.int(len(p)), C.int(params.Gamma()), C.uint64_t(params.T()), C.int(device)))
	runtime.SetFinalizer(ctx, func(c *Context) { C.mkhe_ctx_destroy(c.c) })
	return ctx
}

// BFVRelinKeys holds the device copies of rlkSet.Value[id].Value[0/1] (mkbfv/keys.go:6-9).
type BFVRelinKeys struct {
	B1, B2, D1, D2, V map[string]*SwitchingKey
}

// NewBFVRelinKeys returns an empty key table.
func NewBFVRelinKeys() *BFVRelinKeys {
	return &BFVRelinKeys{B1: map[string]*SwitchingKey{}, B2: map[string]*SwitchingKey{}, D1: map[string]*SwitchingKey{}, D2: map[string]*SwitchingKey{}, V: map[string]*SwitchingKey{}}
}

// UploadBFVRelinKey: party id's key, rlk.Value[g].Value[k].Value for gadget g = 0, 1 and k = 0 (b), 1 (d), 2 (v) (mkbfv/keys.go:6-9:
// two mkrlwe.RelinearizationKeys; v of the second gadget is not used, mkbfv/keygen.go:24-88).
func (ctx *Context) UploadBFVRelinKey(rk *BFVRelinKeys, id string, b1, b2, d1, d2, v []rlwe.PolyQP) {
	rk.B1[id] = ctx.UploadSwitchingKey(b1)
	rk.B2[id] = ctx.UploadSwitchingKey(b2)
	rk.D1[id] = ctx.UploadSwitchingKey(d1)
	rk.D2[id] = ctx.UploadSwitchingKey(d2)
	rk.V[id] = ctx.UploadSwitchingKey(v)
}

func swkList(ids []string, m map[string]*SwitchingKey) **C.mkhe_swk {
	arr := (**C.mkhe_swk)(C.malloc(C.size_t(len(ids)+1) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	s := swkSlice(arr, len(ids)+1)
	s[len(ids)] = nil
	for i, id := range ids {
		k, ok := m[id]
		if !ok {
			panic("cannot GetRelinearizationKey: there is no relinearization key with given id") // mkbfv/keys.go:75-83
		}
		s[i] = k.h
	}
	return arr
}

// MulRelinBFV is the body of mkbfv.Evaluator.mulRelinHoisted (evaluator.go:118-140): op0, op1 and out are device
// ciphertexts whose id order is ids0 / ids1 / their union.
func (ctx *Context) MulRelinBFV(op0, op1 *Ciphertext, ids0, ids1 []string, rk *BFVRelinKeys, crsU *SwitchingKey, out *Ciphertext) {
	b1, b2 := swkList(ids1, rk.B1), swkList(ids1, rk.B2)
	d1, d2, v := swkList(ids0, rk.D1), swkList(ids0, rk.D2), swkList(ids0, rk.V)
	defer func() {
		for _, p := range []**C.mkhe_swk{b1, b2, d1, d2, v} {
			C.free(unsafe.Pointer(p))
		}
	}()
	must(C.mkhe_bfv_mul_relin(ctx.c, op0.h, op1.h, b1, b2, d1, d2, v, crsU.h, out.h))
}

// MulRelinBFVUnhoisted is the body of mkbfv.Evaluator.mulRelin (evaluator.go:95-113) -> KeySwitcher.MulAndRelinBFV
// (keyswitch.go:115-251), the non-hoisted twin: same arguments and the same ciphertext as MulRelinBFV, the reference's own order of
// operations on one pair of pool digit vectors.
func (ctx *Context) MulRelinBFVUnhoisted(op0, op1 *Ciphertext, ids0, ids1 []string, rk *BFVRelinKeys, crsU *SwitchingKey, out *Ciphertext) {
	b1, b2 := swkList(ids1, rk.B1), swkList(ids1, rk.B2)
	d1, d2, v := swkList(ids0, rk.D1), swkList(ids0, rk.D2), swkList(ids0, rk.V)
	defer func() {
		for _, p := range []**C.mkhe_swk{b1, b2, d1, d2, v} {
			C.free(unsafe.Pointer(p))
		}
	}()
	must(C.mkhe_bfv_mul_relin_unhoisted(ctx.c, op0.h, op1.h, b1, b2, d1, d2, v, crsU.h, out.h))
}

// ExternalProductBFV is mkbfv.KeySwitcher.ExternalProductBFV (mkbfv/keyswitch.go:92-114, the non-hoisted form: DecomposeBFV of the
// PolyR operand inside): polyR and out are raw device buffers (mkhe_buf_alloc) holding a PolyR (2l limbs) and a PolyQ (l limbs).
func (ctx *Context) ExternalProductBFV(polyR unsafe.Pointer, bg1, bg2 *SwitchingKey, out unsafe.Pointer) {
	must(C.mkhe_bfv_external_product(ctx.c, polyR, bg1.h, bg2.h, out))
}

// ExternalProductBFVHoisted is mkbfv.KeySwitcher.ExternalProductBFVHoisted (mkbfv/keyswitch_hoisted.go:6-34).
func (ctx *Context) ExternalProductBFVHoisted(ah1, ah2, bg1, bg2 *SwitchingKey, out unsafe.Pointer) {
	must(C.mkhe_bfv_external_product_hoisted(ctx.c, ah1.h, ah2.h, bg1.h, bg2.h, out))
}

// AddBFV / SubBFV: mkbfv.Evaluator.AddNew / SubNew (evaluator.go:44-76)
func (ctx *Context) AddBFV(op0, op1, out *Ciphertext) { must(C.mkhe_ct_add(ctx.c, op0.h, op1.h, out.h)) }
func (ctx *Context) SubBFV(op0, op1, out *Ciphertext) { must(C.mkhe_ct_sub(ctx.c, op0.h, op1.h, out.h)) }

