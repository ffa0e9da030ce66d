// keyswitchgpu.go -- the rest of the KeySwitcher / Evaluator method set of SURVEY.md 8(b) on the cgo boundary: ExternalProduct[Hoisted],
// non-hoisted Rotate, Conjugate, HoistedForm, MulAndRelinHoisted, MultByConst, MulPtxt, Add / Sub.
//
// NOT BUILT OR TESTED IN THIS REPOSITORY (no Go toolchain; see mkrlwegpu.go).  tests/test_go_shim_static.py checks every C.mkhe_* call of
// this package against include/mkhe.h (name, number of arguments, pointer / scalar kind of each one).
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
)

// NewSwitchingKey allocates an uninitialised device switching key / hoisted-digit vector (mkrlwe.NewSwitchingKey, keys.go:245-255:
// always beta(maxLevel) digits): the destination of Decompose / HoistedForm.
func (ctx *Context) NewSwitchingKey() *SwitchingKey {
	out := &SwitchingKey{ctx: ctx}
	must(C.mkhe_swk_create_uninit(ctx.c, &out.h))
	runtime.SetFinalizer(out, func(k *SwitchingKey) { k.Close() })
	return out
}

// NewCiphertext allocates a zeroed device ciphertext over ids at the given level (mkrlwe.NewCiphertext, elements.go:22-36).
func (ctx *Context) NewCiphertext(ids []string, level int) *Ciphertext { return ctx.newCt(ids, level) }

func b2i(b bool) C.int {
	if b {
		return 1
	}
	return 0
}

// swkSlice / ctSlice view n C pointers at arr as a Go slice.  (The array-pointer idiom, not unsafe.Slice: the reference's go.mod says `go 1.13`,
// and unsafe.Slice needs language version 1.17.)
func swkSlice(arr **C.mkhe_swk, n int) []*C.mkhe_swk {
	return (*[1 << 28]*C.mkhe_swk)(unsafe.Pointer(arr))[:n:n]
}

func ctSlice(arr **C.mkhe_ct, n int) []*C.mkhe_ct {
	return (*[1 << 28]*C.mkhe_ct)(unsafe.Pointer(arr))[:n:n]
}

// swkArray copies the handles into C memory (NULL-terminated by one spare slot): a Go slice of C pointers may not be handed to C as **T
// while it holds Go-allocated backing under the cgo pointer rules only if it contains Go pointers -- these are C pointers, but the
// array itself is then pinned for the call by being C memory.  The caller frees it.
func swkArray(keys []*SwitchingKey) **C.mkhe_swk {
	arr := (**C.mkhe_swk)(C.malloc(C.size_t(len(keys)+1) * C.size_t(unsafe.Sizeof(uintptr(0)))))
	s := swkSlice(arr, len(keys)+1)
	for i, k := range keys {
		s[i] = k.h
	}
	s[len(keys)] = nil
	return arr
}

// ExternalProduct replaces KeySwitcher.ExternalProduct (mkrlwe/keyswitch.go:79-118): a = slot `slot` of ct (0 = "0", 1 + i = ids[i]),
// isNTT = a.IsNTT; the result (coefficient domain, canonical) is written to slot outSlot of out, which must have levelQ + 1 limbs.
func (ctx *Context) ExternalProduct(levelQ int, ct *Ciphertext, slot int, isNTT bool, bg *SwitchingKey, out *Ciphertext, outSlot int) {
	must(C.mkhe_external_product(ctx.c, C.int(levelQ), b2i(isNTT), ct.h, C.int(slot), bg.h, out.h, C.int(outSlot)))
}

// ExternalProductHoisted replaces KeySwitcher.ExternalProductHoisted (mkrlwe/keyswitch_hoisted.go:10-40).
func (ctx *Context) ExternalProductHoisted(levelQ int, aHoisted, bg *SwitchingKey, out *Ciphertext, outSlot int) {
	must(C.mkhe_external_product_hoisted(ctx.c, C.int(levelQ), aHoisted.h, bg.h, out.h, C.int(outSlot)))
}

// HoistedForm replaces mkckks.Evaluator.HoistedForm (mkckks/evaluator.go:543-553): one Decompose per party component, as one batched launch.
// The returned keys are aligned with ct.ids.
func (ctx *Context) HoistedForm(ct *Ciphertext, level int) []*SwitchingKey {
	out := make([]*SwitchingKey, len(ct.ids))
	for i := range out {
		out[i] = ctx.NewSwitchingKey()
	}
	arr := swkArray(out)
	defer C.free(unsafe.Pointer(arr))
	must(C.mkhe_hoisted_form(ctx.c, C.int(level), ct.h, arr))
	return out
}

// MulAndRelinHoisted replaces KeySwitcher.MulAndRelinHoisted (mkrlwe/keyswitch_hoisted.go:44-179): hoisted0 / hoisted1 aligned with the ids
// of op0 / op1 (HoistedForm), or nil (the engine hoists internally, = MulAndRelin).
func (ctx *Context) MulAndRelinHoisted(op0, op1 *Ciphertext, hoisted0, hoisted1 []*SwitchingKey, rk RelinKeys, crsU *SwitchingKey, out *Ciphertext) {
	b1 := handles(op1.ids, rk, 0)
	d0 := handles(op0.ids, rk, 1)
	v0 := handles(op0.ids, rk, 2)
	var h0, h1 **C.mkhe_swk
	if hoisted0 != nil {
		h0 = swkArray(hoisted0)
		defer C.free(unsafe.Pointer(h0))
	}
	if hoisted1 != nil {
		h1 = swkArray(hoisted1)
		defer C.free(unsafe.Pointer(h1))
	}
	must(C.mkhe_mul_and_relin(ctx.c, op0.h, op1.h, h0, h1,
		(**C.mkhe_swk)(unsafe.Pointer(&b1[0])), (**C.mkhe_swk)(unsafe.Pointer(&d0[0])),
		(**C.mkhe_swk)(unsafe.Pointer(&v0[0])), crsU.h, out.h))
}

// Rotate replaces the non-hoisted KeySwitcher.Rotate (mkrlwe/keyswitch.go:234-298): rk aligned with in.ids (rkSet.GetRotationKey(id, rotidx)),
// crs = params.CRS[rotidx].  The engine decomposes the party components itself, like the reference's ExternalProduct calls.
func (ctx *Context) Rotate(in *Ciphertext, rotidx int, rk []*SwitchingKey, crs *SwitchingKey, out *Ciphertext) {
	ctx.RotateHoisted(in, rotidx, nil, rk, crs, out)
}

// Conjugate replaces KeySwitcher.Conjugate (mkrlwe/keyswitch.go:302-332): ck aligned with in.ids (ckSet.GetConjugationKey(id)), crs = params.CRS[-2].
func (ctx *Context) Conjugate(in *Ciphertext, ck []*SwitchingKey, crs *SwitchingKey, out *Ciphertext) {
	galEl := ctx.params.GaloisElementForRowRotation()
	arr := swkArray(ck)
	defer C.free(unsafe.Pointer(arr))
	must(C.mkhe_conjugate(ctx.c, C.uint64_t(galEl), in.h, arr, crs.h, out.h))
}

// Add / Sub replace mkckks.Evaluator.AddNew / SubNew (mkckks/evaluator.go:316-356): out carries the union of the id sets.
func (ctx *Context) Add(op0, op1, out *Ciphertext) { must(C.mkhe_ct_add(ctx.c, op0.h, op1.h, out.h)) }
func (ctx *Context) Sub(op0, op1, out *Ciphertext) { must(C.mkhe_ct_sub(ctx.c, op0.h, op1.h, out.h)) }

// MultByConst is the coefficient loop of mkckks.Evaluator.MultByConst (mkckks/evaluator.go:117-199).  scaledReal[i] / scaledImag[i] are the
// reference's scaleUpExact(cReal, scale, q_i) / scaleUpExact(cImag, scale, q_i) (0 where the part is zero): getConstAndScale and scaleUpExact
// (:40-94) stay in Go, as does ctOut.Scale = ct0.Scale * scale.  The two Montgomery-form constants per limb are formed exactly as at :133-150,172-175.
func (ctx *Context) MultByConst(in *Ciphertext, scaledReal, scaledImag []uint64, out *Ciphertext) {
	ringQ := ctx.params.RingQ()
	n := len(scaledReal)
	first := make([]uint64, n)
	second := make([]uint64, n)
	for i := 0; i < n; i++ {
		qi := ringQ.Modulus[i]
		re, im := scaledReal[i], uint64(0)
		c0, c1 := re, re
		if scaledImag[i] != 0 {
			im = ring.MRed(scaledImag[i], ringQ.NttPsi[i][1], qi, ringQ.MredParams[i])
			c0 = ring.CRed(re+im, qi)
			c1 = ring.CRed(re+(qi-im), qi)
		}
		first[i] = ring.MForm(c0, qi, ringQ.BredParams[i])
		second[i] = ring.MForm(c1, qi, ringQ.BredParams[i])
	}
	must(C.mkhe_ct_mul_const(ctx.c, in.h, (*C.uint64_t)(unsafe.Pointer(&first[0])), (*C.uint64_t)(unsafe.Pointer(&second[0])), out.h))
	runtime.KeepAlive(first)
	runtime.KeepAlive(second)
}

// Plaintext is a device copy of a ckks.Plaintext polynomial (coefficient domain), uint64[limbs][N].
type Plaintext struct {
	dev unsafe.Pointer
	ctx *Context
}

func (p *Plaintext) Close() {
	if p.dev != nil {
		C.mkhe_buf_free(p.ctx.c, p.dev)
		p.dev = nil
	}
	runtime.SetFinalizer(p, nil)
}

// UploadPlaintext copies pt.Value (level + 1 limbs) to the device.
func (ctx *Context) UploadPlaintext(pt *ring.Poly, level int) *Plaintext {
	n := ctx.params.N()
	buf := make([]uint64, (level+1)*n)
	for j := 0; j <= level; j++ {
		copy(buf[j*n:(j+1)*n], pt.Coeffs[j])
	}
	out := &Plaintext{ctx: ctx}
	must(C.mkhe_buf_alloc(ctx.c, C.size_t(len(buf)), &out.dev))
	runtime.SetFinalizer(out, func(p *Plaintext) { p.Close() })
	must(C.mkhe_buf_upload(ctx.c, out.dev, (*C.uint64_t)(unsafe.Pointer(&buf[0])), C.size_t(len(buf))))
	runtime.KeepAlive(buf)
	return out
}

// MulPtxt is the body of mkckks.Evaluator.MulPtxtNew without its Rescale (mkckks/evaluator.go:465-478): every component times the plaintext
// (NTT, MForm, product, InvNTT on the device); the caller follows with Rescale, as the reference does at :480.
func (ctx *Context) MulPtxt(in *Ciphertext, pt *Plaintext, out *Ciphertext) {
	must(C.mkhe_ct_mul_ptxt(ctx.c, in.h, pt.dev, out.h))
}

// Sync waits for everything enqueued on the context (the reference is synchronous; Download synchronises by itself).
func (ctx *Context) Sync() { must(C.mkhe_ctx_sync(ctx.c)) }

