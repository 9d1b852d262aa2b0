"""Minimal WebAssembly (MVP) binary reader: sections, function names, data segments and the
i32.const immediates of each function body.  Own code; used only by fixture/constant generators
in this container (the wasm artifact does not travel to the GPU box)."""
import struct

def uleb(b, p):
    r = 0; s = 0
    while True:
        x = b[p]; p += 1
        r |= (x & 0x7F) << s; s += 7
        if not x & 0x80: return r, p

def sleb(b, p, bits=64):
    r = 0; s = 0
    while True:
        x = b[p]; p += 1
        r |= (x & 0x7F) << s; s += 7
        if not x & 0x80:
            if x & 0x40: r -= 1 << s
            return r, p

class Wasm:
    def __init__(self, path):
        self.b = b = open(path, 'rb').read()
        assert b[:4] == b'\0asm'
        p = 8
        self.sections = {}
        self.custom = {}
        while p < len(b):
            sid = b[p]; p += 1
            size, p = uleb(b, p)
            if sid == 0:
                nl, q = uleb(b, p); name = b[q:q+nl].decode(); self.custom[name] = (q+nl, p+size)
            else:
                self.sections[sid] = (p, p+size)
            p += size
        self._imports(); self._exports(); self._code(); self._data(); self._names()

    def _imports(self):
        self.n_imported_funcs = 0
        if 2 not in self.sections: return
        p, e = self.sections[2]; b = self.b
        n, p = uleb(b, p)
        for _ in range(n):
            l, p = uleb(b, p); p += l
            l, p = uleb(b, p); p += l
            kind = b[p]; p += 1
            if kind == 0: _, p = uleb(b, p); self.n_imported_funcs += 1
            elif kind == 1: p += 1; f = b[p]; p += 1; _, p = uleb(b, p); p = uleb(b, p)[1] if f & 1 else p
            elif kind == 2: f = b[p]; p += 1; _, p = uleb(b, p); p = uleb(b, p)[1] if f & 1 else p
            elif kind == 3: p += 2

    def _exports(self):
        self.exports = {}
        p, e = self.sections[7]; b = self.b
        n, p = uleb(b, p)
        for _ in range(n):
            l, p = uleb(b, p); name = b[p:p+l].decode(); p += l
            kind = b[p]; p += 1
            idx, p = uleb(b, p)
            self.exports[name] = (kind, idx)

    def _code(self):
        self.bodies = []
        p, e = self.sections[10]; b = self.b
        n, p = uleb(b, p)
        for _ in range(n):
            size, p = uleb(b, p)
            self.bodies.append((p, p+size)); p += size

    def _data(self):
        self.data = []
        p, e = self.sections[11]; b = self.b
        n, p = uleb(b, p)
        for _ in range(n):
            flag, p = uleb(b, p)
            assert flag == 0
            assert b[p] == 0x41
            off, p = sleb(b, p+1)
            assert b[p] == 0x0B; p += 1
            size, p = uleb(b, p)
            self.data.append((off, p, size)); p += size

    def _names(self):
        self.func_names = {}
        if 'name' not in self.custom: return
        p, e = self.custom['name']; b = self.b
        while p < e:
            sub = b[p]; p += 1
            size, p = uleb(b, p)
            if sub == 1:
                q = p; n, q = uleb(b, q)
                for _ in range(n):
                    idx, q = uleb(b, q); l, q = uleb(b, q)
                    self.func_names[idx] = b[q:q+l].decode(); q += l
            p += size

    def mem_read(self, addr, n):
        for off, fp, size in self.data:
            if off <= addr and addr + n <= off + size:
                return self.b[fp + addr - off: fp + addr - off + n]
        raise KeyError(addr)

    def i32_consts(self, func_index):
        """All i32.const immediates of a function body, in program order (proper MVP decode)."""
        b = self.b
        p, e = self.bodies[func_index - self.n_imported_funcs]
        nl, p = uleb(b, p)
        for _ in range(nl):
            _, p = uleb(b, p); p += 1
        out = []
        while p < e:
            op = b[p]; p += 1
            if op in (0x02, 0x03, 0x04):            # block/loop/if blocktype
                _, p = sleb(b, p)
            elif op in (0x0C, 0x0D, 0x10, 0x20, 0x21, 0x22, 0x23, 0x24):
                _, p = uleb(b, p)
            elif op == 0x0E:
                n, p = uleb(b, p)
                for _ in range(n + 1): _, p = uleb(b, p)
            elif op == 0x11:
                _, p = uleb(b, p); _, p = uleb(b, p)
            elif 0x28 <= op <= 0x3E:
                _, p = uleb(b, p); _, p = uleb(b, p)
            elif op in (0x3F, 0x40):
                p += 1
            elif op == 0x41:
                v, p = sleb(b, p); out.append(v)
            elif op == 0x42:
                _, p = sleb(b, p)
            elif op == 0x43: p += 4
            elif op == 0x44: p += 8
            elif op == 0xFC:
                sub, p = uleb(b, p)
                if sub in (8,): _, p = uleb(b, p); p += 1
                elif sub in (9,): _, p = uleb(b, p)
                elif sub in (10,): p += 2
                elif sub in (11,): p += 1
            # all other MVP opcodes have no immediates
        return out
