// keygengpu.go -- cgo binding of the mkhe_keygen_* / mkhe_crs_expand entry points (include/mkhe.h) for
// mkrlwe.KeyGenerator (mkrlwe/keygen.go) and the CRS slots of mkrlwe.Parameters (params.go:37-61,77-99).
//
// NOT BUILT OR TESTED IN THIS REPOSITORY (no Go toolchain; see mkrlwegpu.go).  The secret and error SAMPLES stay
// with lattigo's samplers (crypto PRNG, utils.NewPRNG): the shim reads them out of a sampled ring.Poly as small
// signed integers and hands those to the engine, which does the NTTs and the products with the CRS on the GPU.
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
	"unsafe"

	"github.com/ldsec/lattigo/v2/ring"
)

// SecretKey is a device PolyQP (NTT, Montgomery form): mkrlwe.SecretKey.Value (keys.go:9-12).
type SecretKey struct {
	ID string
	d  unsafe.Pointer
}

// smallCoeffs turns what a lattigo sampler wrote under modulus q_0 (s or q_0 - |s|) back into signed integers,
// the same centring ExtendBasisSmallNormAndCenter applies (keygen.go:51,130).
func smallCoeffs(dst []C.int32_t, limb0 []uint64, q0 uint64) {
	for i, c := range limb0 {
		if c > q0>>1 {
			dst[i] = -C.int32_t(q0 - c)
		} else {
			dst[i] = C.int32_t(c)
		}
	}
}

// sampleErrors draws `count` Gaussian error polynomials with the reference's sampler (keygen.go:36,124-134).
func (ctx *Context) sampleErrors(g *ring.GaussianSampler, count int) []C.int32_t {
	rq := ctx.params.RingQ()
	tmp := rq.NewPoly()
	out := make([]C.int32_t, count*rq.N)
	for k := 0; k < count; k++ {
		g.ReadLvl(0, tmp)
		smallCoeffs(out[k*rq.N:(k+1)*rq.N], tmp.Coeffs[0], rq.Modulus[0])
	}
	return out
}

// GenSecretKey is genSecretKeyFromSampler (keygen.go:44-55) with the NTT / MForm on the device.
func (ctx *Context) GenSecretKey(sampler ring.Sampler, id string) *SecretKey {
	rq := ctx.params.RingQ()
	tmp := rq.NewPoly()
	sampler.Read(tmp)
	s := make([]C.int32_t, rq.N)
	smallCoeffs(s, tmp.Coeffs[0], rq.Modulus[0])
	sk := &SecretKey{ID: id}
	words := C.size_t((ctx.params.QCount() + ctx.params.PCount()) * rq.N)
	must(C.mkhe_buf_alloc(ctx.c, words, &sk.d))
	must(C.mkhe_keygen_secret(ctx.c, &s[0], sk.d))
	return sk
}

// ExpandCRS replaces the upload of params.CRS[idx] (56 MiB at PN15QP880) by its expansion from a public seed.
func (ctx *Context) ExpandCRS(seed uint64, idx int) *SwitchingKey {
	out := &SwitchingKey{ctx: ctx}
	must(C.mkhe_swk_create(ctx.c, &out.h))
	must(C.mkhe_crs_expand(ctx.c, C.uint64_t(seed), C.int32_t(idx), out.h))
	return out
}

// GenRelinearizationKey is keygen.go:137-187: (b, d, v) stay on the device, ready for MulAndRelin.
func (ctx *Context) GenRelinearizationKey(g *ring.GaussianSampler, sk, r *SecretKey, a, u *SwitchingKey) (b, d, v *SwitchingKey) {
	beta := ctx.params.Beta(ctx.params.MaxLevel())
	e := ctx.sampleErrors(g, 3*beta)
	b, d, v = &SwitchingKey{ctx: ctx}, &SwitchingKey{ctx: ctx}, &SwitchingKey{ctx: ctx}
	for _, k := range []*SwitchingKey{b, d, v} {
		must(C.mkhe_swk_create(ctx.c, &k.h))
	}
	must(C.mkhe_keygen_relin_key(ctx.c, sk.d, r.d, &e[0], a.h, u.h, b.h, d.h, v.h))
	return
}

// GenRotationKey is keygen.go:190-229 (crs = CRS[rotidx]).
func (ctx *Context) GenRotationKey(g *ring.GaussianSampler, rotidx int, sk *SecretKey, crs *SwitchingKey) *SwitchingKey {
	for rotidx < 0 {
		rotidx += ctx.params.N() / 2
	}
	e := ctx.sampleErrors(g, ctx.params.Beta(ctx.params.MaxLevel()))
	out := &SwitchingKey{ctx: ctx}
	must(C.mkhe_swk_create(ctx.c, &out.h))
	must(C.mkhe_keygen_rotation_key(ctx.c, C.uint64_t(ctx.params.GaloisElementForColumnRotationBy(rotidx)), sk.d, &e[0], crs.h, out.h))
	return out
}
