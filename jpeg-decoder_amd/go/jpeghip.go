// jpeghip.go -- cgo shim that puts libjpeghip.so behind maxasm/jpeg-decoder's Go host.
//
// Drop this file next to main.go (package main).  It replaces the two call sites of decodeStartOfScan
// that SURVEY.md section 8b names:
//
//	cut 1  main.go:1022-1032  generateCodes + decodeHuffmanData(header, &BitReader{...})
//	cut 2  main.go:1054-1057  dequantize; inverseDCT; spreadCoeffecients; convertColorSpace
//
// with ONE call, decodeOnGPU(header, rawECS), that returns the cropped RGB8 image writeBitMap needs.
// The Go host keeps every marker decoder unchanged; the only host change is that the ECS copy loop
// (main.go:993-1021) must hand over the RAW stuffed bytes (FF00 and RSTn left in place) because the
// reference's loop throws the restart markers away (SURVEY F3).  See INTEGRATION.md for the patch.
//
// NOT COMPILED IN THIS REPOSITORY'S IMAGE: there is no Go toolchain here (`go version`: command not
// found).  The C side of every call below is exercised by tests/ through ctypes with the same
// argument marshalling.
//
// Build:  CGO_ENABLED=1 CGO_CFLAGS="-I<repo>/include" CGO_LDFLAGS="-L<repo>/jpeg-decoder_amd -ljpeghip" go build .
package main

/*
#include <stdlib.h>
#include <string.h>
#include "jpeghip.h"
*/
import "C"

import (
	"fmt"
	"os"
	"unsafe"
)

var jdReady bool

// jdInit binds the library to every visible MI355X once (jd_init is idempotent).
func jdInit() {
	if jdReady {
		return
	}
	if rc := C.jd_init(nil, 0); rc != 0 {
		fmt.Printf("Error! %s\n", C.GoString(C.jd_strerror(rc)))
		os.Exit(1)
	}
	jdReady = true
}

// frameFromHeader marshals what Header holds at cut 1 (main.go:1467-1491) into a jd_frame.
func frameFromHeader(header *Header) C.jd_frame {
	var f C.jd_frame
	f.width = C.uint16_t(header.width)
	f.height = C.uint16_t(header.height)
	f.ncomp = C.uint8_t(len(header.cComponents))
	f.restart_interval = C.uint16_t(header.restartInterval)
	for i := range header.cComponents {
		c := &header.cComponents[i]
		f.comp[i].h = C.uint8_t(c.hSamplingFactor)
		f.comp[i].v = C.uint8_t(c.vSamplingFactor)
		f.comp[i].tq = C.uint8_t(c.qTableId)
		f.comp[i].td = C.uint8_t(c.dcHuffmanTableId)
		f.comp[i].ta = C.uint8_t(c.acHuffmanTableId)
	}
	for i := range header.qTables { // natural order bytes, exactly as main.go:61-75 stores them
		q := &header.qTables[i]
		for k := 0; k < 64; k++ {
			f.qt[q.Id][k] = C.uint8_t(q.table[k])
		}
		f.qt_present |= 1 << uint(q.Id)
	}
	for i := range header.huffmanTables { // codesOfLen + symbols (main.go:853-870); codes[] is not needed
		t := &header.huffmanTables[i]
		cls := 1
		if t.dc {
			cls = 0
		}
		h := &f.huff[cls][t.Id]
		for k := 0; k < 16; k++ {
			h.bits[k] = C.uint8_t(t.codesOfLen[k])
		}
		for k := range t.symbols {
			h.vals[k] = C.uint8_t(t.symbols[k])
		}
		h.nvals = C.uint16_t(len(t.symbols))
		h.present = 1
	}
	return f
}

// decodeOnGPU replaces main.go:1022-1032 and 1054-1057.  rawECS is the stuffed entropy-coded segment
// between the SOS header and EOI.  The result is H x W x 3 bytes, top-down, R,G,B.
func decodeOnGPU(header *Header, rawECS []byte) []byte {
	jdInit()
	f := frameFromHeader(header)
	if rc := C.jd_frame_check(&f); rc != 0 {
		fmt.Printf("%s\n", C.GoString(C.jd_strerror(rc)))
		os.Exit(1)
	}
	rgb := make([]byte, header.width*header.height*3)
	// cgo rule: C must not keep Go pointers after the call returns -- jd_decode is synchronous and
	// copies the ECS into device memory before it returns.
	rc := C.jd_decode(&f, (*C.uint8_t)(unsafe.Pointer(&rawECS[0])), C.size_t(len(rawECS)), 1, nil, 0,
		(*C.uint8_t)(unsafe.Pointer(&rgb[0])), C.size_t(header.width*3), 0)
	if rc != 0 {
		// the reference prints and exits 1 on every decode error (main.go:189-192, 1014-1016)
		fmt.Printf("%s\n", C.GoString(C.jd_strerror(rc)))
		os.Exit(1)
	}
	return rgb
}

// blocksFromRGB refills Header.blocks so the UNMODIFIED writeBitMap (main.go:1171-1218) can be kept:
// ch1=R, ch2=G, ch3=B at block (x/8, y/8), element (x%8)+(y%8)*8 (main.go:1199-1211).
func blocksFromRGB(header *Header, rgb []byte) {
	blocks := *header.blocks
	for y := 0; y < header.height; y++ {
		for x := 0; x < header.width; x++ {
			b := &blocks[(x/8)+(y/8)*header.blockWidthReal]
			pi := (x % 8) + (y%8)*8
			o := (y*header.width + x) * 3
			b.ch1[pi], b.ch2[pi], b.ch3[pi] = int(rgb[o]), int(rgb[o+1]), int(rgb[o+2])
		}
	}
}

// writeBitMapRGB is the buffered form of writeBitMap fed straight from RGB8 (SURVEY row N1): the
// library lays the file out byte-for-byte as main.go:1171-1218 does.
func writeBitMapRGB(name string, rgb []byte, width, height int) {
	n := C.jd_bmp_encode(nil, 0, C.int(width), C.int(height), nil, 0)
	out := make([]byte, int(n))
	C.jd_bmp_encode((*C.uint8_t)(unsafe.Pointer(&rgb[0])), C.size_t(width*3), C.int(width), C.int(height),
		(*C.uint8_t)(unsafe.Pointer(&out[0])), C.size_t(n))
	if err := os.WriteFile(name, out, 0o644); err != nil {
		fmt.Printf("Error! %v\n", err)
		os.Exit(1)
	}
}
