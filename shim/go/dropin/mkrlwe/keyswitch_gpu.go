// keyswitch_gpu.go -- mkrlwe.KeySwitcher on an MI355X: the SAME type name, constructor and method set as mkrlwe/keyswitch.go and
// mkrlwe/keyswitch_hoisted.go, on the reference's own host types, with the polynomial work done by libmkhe_hip.so (include/mkhe.h) through
// mk-lattigo/mkrlwegpu.  NOT BUILT OR TESTED IN THIS REPOSITORY (no Go toolchain, lattigo v2.3.0 not vendored); tests/test_go_dropin_static.py
// holds every exported signature below against tests/golden/ref_go_signatures.json (the reference's, read from its sources).
//
// Installation (shim/go/install.sh does it): copy this file into the reference's mkrlwe/ directory, copy shim/go/mkrlwegpu into the module
// root, and apply shim/go/patches/mkrlwe_build_tags.diff, which puts `//go:build !mkhe_gpu` on keyswitch.go and keyswitch_hoisted.go.  Then
//
//	go test -tags mkhe_gpu ./mkrlwe ./mkckks ./mkbfv        (CGO_CFLAGS / CGO_LDFLAGS as in mkrlwegpu.go)
//
// runs the reference's own tests -- mkrlwe_test.go:456-679 (ExternalProduct, Decompose), mkckks_test.go:320-362, mkbfv_test.go:282-416 -- through
// the engine with NO change to them: mkckks.Evaluator and mkbfv.KeySwitcher reach the polynomial layer only through this type
// (mkckks/evaluator.go:423-612, mkbfv/keyswitch.go:32-33,65).  Why a file of package mkrlwe and not a type of package mkrlwegpu: the tests are
// in-package (`package mkrlwe`), and Go does not let a test of a package import something that imports the package.
//
// Residency (the engine keeps operands in HBM; the reference passes host structs):
//   - keys and CRS (*SwitchingKey reached through rlkSet / rkSet / ckSet / Parameters.CRS): uploaded at their first use, found by pointer
//     afterwards.  A key that is MODIFIED in place after its first use is noticed: every lookup compares a fingerprint of the host struct
//     (64 words sampled across its digits and limbs, taken at the upload) and uploads again on a mismatch; MKHE_GO_TRUST_RESIDENT=1 skips the
//     comparison (a caller that never rewrites a key saves 64 loads per key and call), Forget drops a copy explicitly.
//   - hoisted digit vectors written by Decompose: the device copy is bound to the host *SwitchingKey it was written "into" and is what
//     ExternalProductHoisted / MulAndRelinHoisted / RotateHoisted read.  HostMirror (default true) also writes the digits into the host
//     struct, as the reference does -- 56 MiB over PCIe per Decompose at PN15QP880; callers that only pass hoisted forms on (mkckks.Evaluator
//     does: evaluator.go:423-437,549,579) set it to false, and Materialize fetches a vector if it is ever wanted on the host.
//   - ciphertext polynomials: uploaded and downloaded per call.
//   - device copies are released when the host struct they are bound to is collected (a finalizer on the *SwitchingKey queues them; the next
//     engine call of the owner closes them) or by Forget.
//
//go:build mkhe_gpu
// +build mkhe_gpu

package mkrlwe

import (
	"os"
	"runtime"
	"strconv"
	"sync"
	"unsafe"

	"github.com/ldsec/lattigo/v2/ring"
	"github.com/ldsec/lattigo/v2/rlwe"

	"mk-lattigo/mkrlwegpu"
)

// KeySwitcher is a struct for RLWE key-switching (mkrlwe/keyswitch.go:8-16).  The embedded rlwe.KeySwitcher and the Decomposer stay: mkbfv's
// host code reaches into them (ks.Pool, ks.Baseconverter: mkbfv/keyswitch.go:92,110, keyswitch_hoisted.go:16,34).
type KeySwitcher struct {
	rlwe.KeySwitcher
	Parameters
	Decomposer *Decomposer

	// HostMirror: Decompose also writes its digits into the host SwitchingKey (see the file comment).
	HostMirror bool
	// Device is the HIP device of the engine context (MKHE_GO_DEVICE, default 0); read when the context is created, at the first device call.
	Device int

	gpu      *mkrlwegpu.Context
	mu       sync.Mutex
	resident map[uintptr]*mkrlwegpu.SwitchingKey // host *SwitchingKey (as an address: no strong reference) -> device copy
	prints   map[uintptr]uint64                  // ... -> fingerprint of the host words the copy was uploaded from (absent: a device-born vector, Adopt)
	trust    bool                                // MKHE_GO_TRUST_RESIDENT=1: no fingerprint comparison at a lookup
	garbage  []*mkrlwegpu.SwitchingKey          // device copies whose host struct was collected: closed by the next engine call (finalizers run on
	// their own goroutine, and the calls on one engine context are serialized by its owner)
}

func NewKeySwitcher(params Parameters) *KeySwitcher {
	ks := new(KeySwitcher)
	ks.KeySwitcher = *rlwe.NewKeySwitcher(params.Parameters)
	ks.Parameters = params
	ks.Decomposer = NewDecomposer(params.RingQ(), params.RingP(), params.Gamma())
	ks.HostMirror = true
	if v, err := strconv.Atoi(os.Getenv("MKHE_GO_DEVICE")); err == nil {
		ks.Device = v
	}
	ks.resident = make(map[uintptr]*mkrlwegpu.SwitchingKey)
	ks.prints = make(map[uintptr]uint64)
	ks.trust = os.Getenv("MKHE_GO_TRUST_RESIDENT") == "1"
	return ks
}

// GPU returns the engine context (created at the first use: mkbfv holds a second KeySwitcher over the ring R that only ever decomposes on the host).
func (ks *KeySwitcher) GPU() *mkrlwegpu.Context {
	if ks.gpu == nil {
		ks.gpu = mkrlwegpu.NewContext(ks.Parameters, ks.Device)
	}
	ks.mu.Lock()
	dead := ks.garbage
	ks.garbage = nil
	ks.mu.Unlock()
	for _, d := range dead {
		d.Close()
	}
	return ks.gpu
}

// collected is the finalizer of a host SwitchingKey that has a device copy: the copy is handed to the next engine call for closing.
func (ks *KeySwitcher) collected(swk *SwitchingKey) {
	key := uintptr(unsafe.Pointer(swk))
	ks.mu.Lock()
	if d, ok := ks.resident[key]; ok {
		delete(ks.resident, key)
		delete(ks.prints, key)
		ks.garbage = append(ks.garbage, d)
	}
	ks.mu.Unlock()
}

// ---- residency

func (ks *KeySwitcher) bind(swk *SwitchingKey, d *mkrlwegpu.SwitchingKey) {
	key := uintptr(unsafe.Pointer(swk))
	ks.mu.Lock()
	old, had := ks.resident[key]
	ks.resident[key] = d
	ks.mu.Unlock()
	if had && old != d {
		old.Close()
	}
	if !had {
		// the device copy lives as long as the host struct it stands for (a struct that cannot carry a finalizer -- not a heap allocation of its
		// own, or it has one already -- keeps its device copy until Forget)
		func() {
			defer func() { _ = recover() }()
			runtime.SetFinalizer(swk, ks.collected)
		}()
	}
}

func (ks *KeySwitcher) lookup(swk *SwitchingKey) *mkrlwegpu.SwitchingKey {
	ks.mu.Lock()
	d := ks.resident[uintptr(unsafe.Pointer(swk))]
	ks.mu.Unlock()
	return d
}

// fingerprint mixes up to 64 words of a host SwitchingKey, spread over its digits, its Q and P limbs and the coefficients of a limb, into 64
// bits (splitmix64 steps): cheap enough for every lookup, and a key generator that rewrites a key in place (mkrlwe/keygen.go:137-187 fills
// every coefficient of every limb with fresh uniform / error terms) changes all of the sampled words.
func fingerprint(v []rlwe.PolyQP) uint64 {
	h := uint64(0x4D4B4845) ^ uint64(len(v))<<32
	mix := func(w uint64) {
		h += w + 0x9E3779B97F4A7C15
		h = (h ^ (h >> 30)) * 0xBF58476D1CE4E5B9
		h = (h ^ (h >> 27)) * 0x94D049BB133111EB
		h ^= h >> 31
	}
	if len(v) == 0 {
		return h
	}
	for s := 0; s < 64; s++ {
		d := (s * 7) % len(v)
		var limbs [][]uint64
		if s%4 == 3 && v[d].P != nil && len(v[d].P.Coeffs) > 0 {
			limbs = v[d].P.Coeffs
		} else if v[d].Q != nil {
			limbs = v[d].Q.Coeffs
		}
		if len(limbs) == 0 {
			mix(uint64(s))
			continue
		}
		l := limbs[(s*5)%len(limbs)]
		if len(l) == 0 {
			mix(uint64(s))
			continue
		}
		mix(l[(s*2654435761)%len(l)])
	}
	return h
}

// Resident returns the device copy of a key / CRS / hoisted digit vector, uploading swk.Value at its first use -- and again when the host words
// it was uploaded from have changed since (see the file comment; MKHE_GO_TRUST_RESIDENT=1 skips that comparison).
func (ks *KeySwitcher) Resident(swk *SwitchingKey) *mkrlwegpu.SwitchingKey {
	if swk == nil {
		panic("mkrlwe (gpu): nil SwitchingKey")
	}
	key := uintptr(unsafe.Pointer(swk))
	if d := ks.lookup(swk); d != nil {
		if ks.trust {
			return d
		}
		ks.mu.Lock()
		fp, uploaded := ks.prints[key]
		ks.mu.Unlock()
		// (a device-born vector -- Adopt: a hoisted form whose host Value may be empty or a stale mirror -- has no host original to compare with)
		if !uploaded || fp == fingerprint(swk.Value) {
			return d
		}
		// the host struct was rewritten in place after its upload: the copy is stale
	}
	d := ks.GPU().UploadSwitchingKey(swk.Value)
	ks.bind(swk, d)
	ks.mu.Lock()
	ks.prints[key] = fingerprint(swk.Value)
	ks.mu.Unlock()
	return d
}

// Adopt binds a device digit vector (mkrlwegpu.Context.HoistedForm) to a host SwitchingKey whose Value may be empty.
func (ks *KeySwitcher) Adopt(swk *SwitchingKey, d *mkrlwegpu.SwitchingKey) {
	ks.bind(swk, d)
	ks.mu.Lock()
	delete(ks.prints, uintptr(unsafe.Pointer(swk)))
	ks.mu.Unlock()
}

// Forget drops the device copy bound to swk (a key that was regenerated in place; a hoisted form that is no longer needed).
func (ks *KeySwitcher) Forget(swk *SwitchingKey) {
	key := uintptr(unsafe.Pointer(swk))
	ks.mu.Lock()
	d, ok := ks.resident[key]
	delete(ks.resident, key)
	delete(ks.prints, key)
	ks.mu.Unlock()
	if ok {
		d.Close()
	}
}

// Materialize writes the device copy bound to swk into swk.Value (allocating it if it is empty): the host view of a hoisted form that was
// produced with HostMirror off.
func (ks *KeySwitcher) Materialize(swk *SwitchingKey) {
	d := ks.lookup(swk)
	if d == nil {
		return
	}
	if len(swk.Value) == 0 {
		swk.Value = NewSwitchingKey(ks.Parameters).Value
	}
	ks.GPU().DownloadSwitchingKey(d, swk.Value)
}

func (ks *KeySwitcher) hoistedList(h *HoistedCiphertext, ids []string) []*mkrlwegpu.SwitchingKey {
	if h == nil {
		return nil
	}
	out := make([]*mkrlwegpu.SwitchingKey, len(ids))
	for i, id := range ids {
		swk, ok := h.Value[id]
		if !ok {
			panic("mkrlwe (gpu): the hoisted ciphertext lacks id " + id)
		}
		out[i] = ks.Resident(swk)
	}
	return out
}

func (ks *KeySwitcher) relinKeys(rlkSet *RelinearizationKeySet, idLists ...[]string) mkrlwegpu.RelinKeys {
	rk := mkrlwegpu.RelinKeys{}
	for _, ids := range idLists {
		for _, id := range ids {
			if _, ok := rk[id]; ok {
				continue
			}
			rlk := rlkSet.GetRelinearizationKey(id) // panics like the reference when the key is missing (keys.go:190-198)
			rk[id] = [3]*mkrlwegpu.SwitchingKey{ks.Resident(rlk.Value[0]), ks.Resident(rlk.Value[1]), ks.Resident(rlk.Value[2])}
		}
	}
	return rk
}

// ---- the method set of mkrlwe/keyswitch.go and keyswitch_hoisted.go

// DecomposeSingleNTT: one gadget digit (keyswitch.go:21-31).  Host code: its only caller outside this type is mkbfv's DecomposeBFV on the
// (R, P) key switcher, digit by digit into host polynomials (mkbfv/keyswitch.go:65); the device paths decompose whole polynomials.
func (ks *KeySwitcher) DecomposeSingleNTT(levelQ, levelP, alpha, beta, gamma int, c2InvNTT, c2QiQ, c2QiP *ring.Poly) {
	ks.Decomposer.DecomposeAndSplit(levelQ, levelP, alpha, beta, gamma, c2InvNTT, c2QiQ, c2QiP)
	ks.Parameters.RingQ().NTTLvl(levelQ, c2QiQ, c2QiQ)
	ks.Parameters.RingP().NTTLvl(levelP, c2QiP, c2QiP)
}

// Decompose: h(a) in R_QP^beta, NTT domain (keyswitch.go:49-73) -> mkhe_decompose.
func (ks *KeySwitcher) Decompose(levelQ int, a *ring.Poly, ad *SwitchingKey) {
	g := ks.GPU()
	ct := g.UploadPoly(a, levelQ)
	defer ct.Close()
	d := ks.lookup(ad)
	if d == nil {
		d = g.NewSwitchingKey()
		ks.bind(ad, d)
	}
	g.Decompose(levelQ, ct, 0, a.IsNTT, d)
	// (the device copy is the original now, whatever ad.Value held when it was uploaded: no fingerprint to hold it against)
	ks.mu.Lock()
	delete(ks.prints, uintptr(unsafe.Pointer(ad)))
	ks.mu.Unlock()
	if ks.HostMirror {
		g.DownloadSwitchingKey(d, ad.Value)
	}
}

// ExternalProduct: c = ModDown_P(<h(a), bg>), coefficient domain (keyswitch.go:79-118) -> mkhe_external_product.
func (ks *KeySwitcher) ExternalProduct(levelQ int, a *ring.Poly, bg *SwitchingKey, c *ring.Poly) {
	g := ks.GPU()
	ct := g.UploadPoly(a, levelQ)
	defer ct.Close()
	out := g.NewCiphertext(nil, levelQ)
	defer out.Close()
	g.ExternalProduct(levelQ, ct, 0, a.IsNTT, ks.Resident(bg), out, 0)
	g.Download(out, map[string]*ring.Poly{"0": c}, levelQ)
}

// ExternalProductHoisted (keyswitch_hoisted.go:10-40) -> mkhe_external_product_hoisted.
func (ks *KeySwitcher) ExternalProductHoisted(levelQ int, aHoisted, bg *SwitchingKey, c *ring.Poly) {
	g := ks.GPU()
	out := g.NewCiphertext(nil, levelQ)
	defer out.Close()
	g.ExternalProductHoisted(levelQ, ks.Resident(aHoisted), ks.Resident(bg), out, 0)
	g.Download(out, map[string]*ring.Poly{"0": c}, levelQ)
}

// MulAndRelin (keyswitch.go:122-230): the non-hoisted twin, the same ciphertext.
func (ks *KeySwitcher) MulAndRelin(op0, op1 *Ciphertext, rlkSet *RelinearizationKeySet, ctOut *Ciphertext) {
	ks.MulAndRelinHoisted(op0, op1, nil, nil, rlkSet, ctOut)
}

// MulAndRelinHoisted (keyswitch_hoisted.go:44-179) -> mkhe_mul_and_relin.  nil hoisted forms: the engine hoists.
func (ks *KeySwitcher) MulAndRelinHoisted(op0, op1 *Ciphertext, op0Hoisted, op1Hoisted *HoistedCiphertext, rlkSet *RelinearizationKeySet, ctOut *Ciphertext) {
	level := ctOut.Level()
	if op0.Level() < level || op1.Level() < level {
		panic("Cannot MulAndRelin: op0 and op1 have different levels")
	}
	g := ks.GPU()
	d0 := g.Upload(op0.Value, op0.Level())
	defer d0.Close()
	d1 := d0
	if op1 != op0 {
		d1 = g.Upload(op1.Value, op1.Level())
		defer d1.Close()
	}
	out := g.NewCiphertext(mkrlwegpu.SortedIDs(ctOut.Value), level)
	defer out.Close()
	rk := ks.relinKeys(rlkSet, d0.IDs(), d1.IDs())
	g.MulAndRelinHoisted(d0, d1, ks.hoistedList(op0Hoisted, d0.IDs()), ks.hoistedList(op1Hoisted, d1.IDs()), rk, ks.Resident(ks.Parameters.CRS[-1]), out)
	g.Download(out, ctOut.Value, level)
}

func (ks *KeySwitcher) rotationKeys(rkSet *RotationKeySet, ids []string, rotidx int) []*mkrlwegpu.SwitchingKey {
	out := make([]*mkrlwegpu.SwitchingKey, len(ids))
	for i, id := range ids {
		out[i] = ks.Resident(rkSet.GetRotationKey(id, uint(rotidx)).Value)
	}
	return out
}

// Rotate (keyswitch.go:234-298) -> mkhe_rotate, the engine decomposing the party components itself.
func (ks *KeySwitcher) Rotate(ctIn *Ciphertext, rotidx int, rkSet *RotationKeySet, ctOut *Ciphertext) {
	ks.RotateHoisted(ctIn, rotidx, nil, rkSet, ctOut)
}

// RotateHoisted (keyswitch_hoisted.go:183-247) -> mkhe_rotate.
func (ks *KeySwitcher) RotateHoisted(ctIn *Ciphertext, rotidx int, ctInHoisted *HoistedCiphertext, rkSet *RotationKeySet, ctOut *Ciphertext) {
	level := ctOut.Level()
	if ctIn.Level() < level {
		panic("Cannot Rotate: ctIn and ctOut have different levels")
	}
	for rotidx < 0 {
		rotidx += ks.Parameters.N() / 2
	}
	a, ok := ks.Parameters.CRS[rotidx]
	if !ok {
		panic("mkrlwe (gpu): no CRS for rotation index " + strconv.Itoa(rotidx))
	}
	g := ks.GPU()
	in := g.Upload(ctIn.Value, ctIn.Level())
	defer in.Close()
	out := g.NewCiphertext(in.IDs(), level)
	defer out.Close()
	g.RotateHoisted(in, rotidx, ks.hoistedList(ctInHoisted, in.IDs()), ks.rotationKeys(rkSet, in.IDs(), rotidx), ks.Resident(a), out)
	g.Download(out, ctOut.Value, level)
}

// Conjugate (keyswitch.go:302-332) -> mkhe_conjugate.
func (ks *KeySwitcher) Conjugate(ctIn *Ciphertext, ckSet *ConjugationKeySet, ctOut *Ciphertext) {
	level := ctOut.Level()
	if ctIn.Level() < level {
		panic("Cannot Conjugate: ctIn and ctOut have different levels")
	}
	g := ks.GPU()
	in := g.Upload(ctIn.Value, ctIn.Level())
	defer in.Close()
	ck := make([]*mkrlwegpu.SwitchingKey, len(in.IDs()))
	for i, id := range in.IDs() {
		ck[i] = ks.Resident(ckSet.GetConjugationKey(id).Value)
	}
	out := g.NewCiphertext(in.IDs(), level)
	defer out.Close()
	g.Conjugate(in, ck, ks.Resident(ks.Parameters.CRS[-2]), out)
	g.Download(out, ctOut.Value, level)
}
