// dump_test.go -- writes one recorded call of the Go reference in the fixture format of tools/fixture.py.
//
// NOT BUILT OR RUN IN THIS REPOSITORY (no Go toolchain, lattigo not vendored).  Usage for a maintainer with Go:
// copy this file into the reference's mkrlwe/ directory and run
//
//	go test ./mkrlwe -run TestDumpMulAndRelin -args -fixture=/tmp/mr.fix
//
// then, in this repository:   python tests/replay_fixture.py /tmp/mr.fix
// which feeds exactly these bytes to the CPU oracle and to the HIP engine and asserts bit-equality with ctOut
// as computed by mkrlwe.KeySwitcher.MulAndRelin (mkrlwe/keyswitch.go:122-230) -- this pins parity against Go.
package mkrlwe

import (
	"encoding/binary"
	"encoding/json"
	"flag"
	"os"
	"testing"

	"github.com/ldsec/lattigo/v2/ring"
	"github.com/ldsec/lattigo/v2/rlwe"
	"github.com/ldsec/lattigo/v2/utils"
)

var fixturePath = flag.String("fixture", "mr.fix", "output file")

type fxArray struct {
	Name   string `json:"name"`
	Shape  []int  `json:"shape"`
	Offset int    `json:"offset"`
}

type fxWriter struct {
	arrays []fxArray
	words  []uint64
}

func (w *fxWriter) addLimbs(name string, limbs [][]uint64, shape []int) {
	w.arrays = append(w.arrays, fxArray{name, shape, len(w.words)})
	for _, l := range limbs {
		w.words = append(w.words, l...)
	}
}

func (w *fxWriter) addPoly(name string, p *ring.Poly, level int) {
	w.addLimbs(name, p.Coeffs[:level+1], []int{level + 1, len(p.Coeffs[0])})
}

// SwitchingKey -> [beta][nQ+nP][N], Q limbs then P limbs (mkrlwe/keys.go:23-25)
func (w *fxWriter) addSwk(name string, swk *SwitchingKey) {
	var limbs [][]uint64
	for _, v := range swk.Value {
		limbs = append(limbs, v.Q.Coeffs...)
		limbs = append(limbs, v.P.Coeffs...)
	}
	m := len(swk.Value[0].Q.Coeffs) + len(swk.Value[0].P.Coeffs)
	w.addLimbs(name, limbs, []int{len(swk.Value), m, len(swk.Value[0].Q.Coeffs[0])})
}

func (w *fxWriter) write(path string, meta map[string]interface{}) error {
	hdr, err := json.Marshal(map[string]interface{}{"meta": meta, "arrays": w.arrays})
	if err != nil {
		return err
	}
	f, err := os.Create(path)
	if err != nil {
		return err
	}
	defer f.Close()
	f.Write([]byte("MKHEFIX1"))
	binary.Write(f, binary.LittleEndian, uint32(len(hdr)))
	f.Write(hdr)
	pad := (8 - (12+len(hdr))%8) % 8
	f.Write(make([]byte, pad))
	return binary.Write(f, binary.LittleEndian, w.words)
}

func plainPsi(r *ring.Ring) []uint64 {
	out := make([]uint64, len(r.Modulus))
	for i := range out {
		out[i] = ring.InvMForm(r.NttPsi[i][r.N>>1], r.Modulus[i], r.MredParams[i])
	}
	return out
}

func TestDumpMulAndRelin(t *testing.T) {
	lit := rlwe.TestPN13QP218 // any set of TestParams; PN15QP880 for the headline shape
	rp, err := rlwe.NewParametersFromLiteral(lit)
	if err != nil {
		t.Fatal(err)
	}
	params := NewParameters(rp, 2)
	kgen := NewKeyGenerator(params)
	ids := []string{"alice", "bob"}
	level := params.MaxLevel()
	ringQ := params.RingQ()

	prng, _ := utils.NewPRNG()
	sampler := ring.NewUniformSampler(prng, ringQ)

	rlkSet := NewRelinearizationKeyKeySet(params)
	idset := NewIDSet()
	for _, id := range ids {
		sk := kgen.GenSecretKey(id)
		r := kgen.GenSecretKey(id)
		rlkSet.AddRelinearizationKey(kgen.GenRelinearizationKey(sk, r))
		idset.Add(id)
	}
	// arithmetic does not depend on the operands being valid encryptions: uniform polynomials
	op0 := NewCiphertext(params, idset, level)
	op1 := NewCiphertext(params, idset, level)
	for _, ct := range []*Ciphertext{op0, op1} {
		for id := range ct.Value {
			sampler.ReadLvl(level, ct.Value[id])
		}
	}
	out := NewCiphertext(params, idset, level)
	ks := NewKeySwitcher(params)

	w := &fxWriter{}
	w.addSwk("crs_u", params.CRS[-1])
	for _, id := range ids {
		rlk := rlkSet.GetRelinearizationKey(id)
		w.addSwk("rlk/"+id+"/b", rlk.Value[0])
		w.addSwk("rlk/"+id+"/d", rlk.Value[1])
		w.addSwk("rlk/"+id+"/v", rlk.Value[2])
	}
	for id := range op0.Value { // inputs are recorded BEFORE the call
		w.addPoly("op0/"+id, op0.Value[id], level)
		w.addPoly("op1/"+id, op1.Value[id], level)
	}

	ks.MulAndRelin(op0, op1, rlkSet, out)

	for id := range out.Value {
		w.addPoly("out/"+id, out.Value[id], level)
	}
	meta := map[string]interface{}{
		"op": "mkrlwe.MulAndRelin", "logN": params.LogN(), "Q": ringQ.Modulus, "P": params.RingP().Modulus,
		"gamma": params.Gamma(), "psiQ": plainPsi(ringQ), "psiP": plainPsi(params.RingP()),
		"level": level, "ids0": ids, "ids1": ids,
	}
	if err := w.write(*fixturePath, meta); err != nil {
		t.Fatal(err)
	}
}
