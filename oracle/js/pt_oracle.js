// pt_oracle.js -- single-thread Node/JS form of the CPU oracle (TEST INFRASTRUCTURE, not product):
// the same intersect + shade loop as oracle/pt_oracle.cpp in strict binary32 via Math.fround, so a
// "single-thread Node/CPU run of the same loop" (BASELINE.json) exists next to the MI355X numbers.
// tests/test_js_oracle.py checks it bit-for-bit against the C++ oracle; bench.py can time it.
// Cited lines refer to the reference's src/shaders/renderer.wgsl; the bounce / NEE / Russian
// roulette part is the build-defined spec of DESIGN.md section 4.  Node-12-safe CommonJS.
"use strict";
const f = Math.fround;
const INF_T = f(1e30), EPS_TRI = f(1e-7), LEAF = 0x80000000, INVALID = 0xFFFFFFFF, STACK_MAX = 64;

// exact fmaf: a*b is exact in double (48-bit product); the double add may round, so a float
// halfway case is repaired with the add's error term before the final rounding
const _f64 = new Float64Array(1), _u32 = new Uint32Array(_f64.buffer);
function fmaf(a, b, c) {
  const p = a * b, s = p + c;
  if (!isFinite(s)) return f(s);
  const t = s - p, e = (p - (s - t)) + (c - t);
  if (e !== 0) {
    _f64[0] = s;
    if ((_u32[0] & 0x1FFFFFFF) === 0x10000000) {          // s sits exactly on a binary32 rounding boundary
      const up = (e > 0) === (s > 0);                       // true sum is farther from zero than s
      if (up) { _u32[0] += 1; } else { _u32[0] -= 1; }      // nudge one double ulp (low word never wraps here)
      return f(_f64[0]);
    }
  }
  return f(s);
}
const _h = new Float32Array(1), _hu = new Uint32Array(_h.buffer);
function halfToFloat(h) {                                    // exact widening (renderer.wgsl:94-96)
  const s = (h & 0x8000) << 16, mag = h & 0x7fff;
  if (mag >= 0x7c00) { _hu[0] = (s | 0x7f800000 | ((mag & 0x3ff) << 13)) >>> 0; return _h[0]; }
  if (mag >= 0x0400) { _hu[0] = (s | ((mag + (112 << 10)) << 13)) >>> 0; return _h[0]; }
  const v = mag * 5.9604644775390625e-8;
  return s ? -v : v;
}
function dot(ax, ay, az, bx, by, bz) { return f(f(f(ax * bx) + f(ay * by)) + f(az * bz)); }
const wmin = (a, b) => (b < a ? b : a), wmax = (a, b) => (b > a ? b : a);

function mix32(x) { x ^= x >>> 16; x = Math.imul(x, 0x7feb352d); x ^= x >>> 15; x = Math.imul(x, 0x846ca68b); x ^= x >>> 16; return x >>> 0; }
function sampleKey(seed, pixel, sidx) { let h = mix32((seed + 0x9E3779B9) >>> 0); h = mix32((h ^ pixel) >>> 0); return mix32((h ^ sidx) >>> 0); }
function rnd(key, bounce, dim) { const h = mix32((key ^ Math.imul(bounce * 8 + dim + 1, 0x9E3779B1)) >>> 0); return f((h >>> 8) * (1.0 / 16777216.0)); }

function sincos2pi(u, out) {
  const q = f(u * 4), kf = Math.floor(f(q + 0.5)), y = f(f(q - kf) * f(1.57079632679489662)), y2 = f(y * y);
  let sp = fmaf(y2, f(2.7557319e-6), f(-1.9841270e-4)); sp = fmaf(y2, sp, f(8.3333333e-3)); sp = fmaf(y2, sp, f(-1.6666667e-1)); sp = fmaf(y2, sp, 1);
  const sy = f(y * sp);
  let cp = fmaf(y2, f(-2.7557319e-7), f(2.4801587e-5)); cp = fmaf(y2, cp, f(-1.3888889e-3)); cp = fmaf(y2, cp, f(4.1666667e-2)); cp = fmaf(y2, cp, -0.5);
  const cy = fmaf(y2, cp, 1), k = kf & 3;
  if (k === 0) { out[0] = cy; out[1] = sy; } else if (k === 1) { out[0] = -sy; out[1] = cy; } else if (k === 2) { out[0] = -cy; out[1] = -sy; } else { out[0] = sy; out[1] = -cy; }
}

class Oracle {
  // tris: Float32Array(9N), bvh4: Uint32Array(1+8M) in the reference layouts
  // spheres (optional Float32Array(4S), x y z r) switches to the brute-force scene of config C1 (no BVH)
  constructor(tris, bvh4, numTris, spheres) {
    this.tris = tris; this.bvh = bvh4 || new Uint32Array([0]); this.numTris = numTris; this.numNodes = this.bvh[0];
    this.spheres = spheres || null; this.brute = !!spheres;
    this.stack = new Uint32Array(STACK_MAX);
    this.stats = { raysClosest: 0, raysShadow: 0, nodesExamined: 0, trisTested: 0, samples: 0 };
    this.hit = { t: INF_T, tri: INVALID, nx: 0, ny: 0, nz: 0 };
    const l = f(1 / f(Math.sqrt(f(f(f(1 * 1) + f(1.5 * 1.5)) + f(1 * 1)))));      // normalize(1,1.5,1), :349
    this.L = [f(1 * l), f(1.5 * l), f(1 * l)];
    this._sc = [0, 0];
  }

  // slab test of node `i` (renderer.wgsl:121-169 for one lane); returns tmin or NaN when missed
  slab(i, ox, oy, oz, ix, iy, iz, best) {
    const b = this.bvh, base = 1 + i * 8, a = b[base], c = b[base + 1], e = b[base + 2];
    const mnx = halfToFloat(a & 0xffff), mny = halfToFloat(a >>> 16), mnz = halfToFloat(c & 0xffff);
    const mxx = halfToFloat(c >>> 16), mxy = halfToFloat(e & 0xffff), mxz = halfToFloat(e >>> 16);
    if (mnx > mxx || mny > mxy || mnz > mxz) return NaN;
    const t1x = f(f(mnx - ox) * ix), t1y = f(f(mny - oy) * iy), t1z = f(f(mnz - oz) * iz);
    const t2x = f(f(mxx - ox) * ix), t2y = f(f(mxy - oy) * iy), t2z = f(f(mxz - oz) * iz);
    const tmin = wmax(wmax(wmin(t1x, t2x), wmin(t1y, t2y)), wmin(t1z, t2z));
    const tmax = wmin(wmin(wmax(t1x, t2x), wmax(t1y, t2y)), wmax(t1z, t2z));
    return (tmax >= wmax(tmin, 0) && tmin < best) ? tmin : NaN;
  }

  // traverseBVH4Packet with one active lane (renderer.wgsl:210-346), incl. the re-test at pop,
  // the nearest-child swap and the silent push drop; anyhit stops at the first accepted hit
  // BUILD-DEFINED: every triangle (reference Moller-Trumbore) then every sphere, strict t < best
  bruteTrace(ox, oy, oz, dx, dy, dz, anyhit) {
    const H = this.hit, T = this.tris, st = this.stats; H.t = INF_T; H.tri = INVALID;
    for (let ti = 0; ti < this.numTris; ti++) {
      const o = ti * 9; st.trisTested++;
      const v0x = T[o], v0y = T[o + 1], v0z = T[o + 2];
      const e1x = f(T[o + 3] - v0x), e1y = f(T[o + 4] - v0y), e1z = f(T[o + 5] - v0z);
      const e2x = f(T[o + 6] - v0x), e2y = f(T[o + 7] - v0y), e2z = f(T[o + 8] - v0z);
      const px = f(f(dy * e2z) - f(dz * e2y)), py = f(f(dz * e2x) - f(dx * e2z)), pz = f(f(dx * e2y) - f(dy * e2x));
      const det = dot(e1x, e1y, e1z, px, py, pz);
      if (Math.abs(det) < EPS_TRI) continue;
      const inv = f(1 / det), sx = f(ox - v0x), sy = f(oy - v0y), sz = f(oz - v0z);
      const u = f(inv * dot(sx, sy, sz, px, py, pz)); if (u < 0 || u > 1) continue;
      const qx = f(f(sy * e1z) - f(sz * e1y)), qy = f(f(sz * e1x) - f(sx * e1z)), qz = f(f(sx * e1y) - f(sy * e1x));
      const v = f(inv * dot(dx, dy, dz, qx, qy, qz)); if (v < 0 || f(u + v) > 1) continue;
      const t = f(inv * dot(e2x, e2y, e2z, qx, qy, qz));
      if (t > EPS_TRI && t < H.t) {
        H.t = t; H.tri = ti;
        const cx = f(f(e1y * e2z) - f(e1z * e2y)), cy = f(f(e1z * e2x) - f(e1x * e2z)), cz = f(f(e1x * e2y) - f(e1y * e2x));
        const il = f(1 / f(Math.sqrt(dot(cx, cy, cz, cx, cy, cz))));
        H.nx = f(cx * il); H.ny = f(cy * il); H.nz = f(cz * il);
        if (anyhit) return true;
      }
    }
    const S = this.spheres;
    for (let si = 0; si * 4 < S.length; si++) {
      const cx = S[si * 4], cy = S[si * 4 + 1], cz = S[si * 4 + 2], r = S[si * 4 + 3];
      const ocx = f(ox - cx), ocy = f(oy - cy), ocz = f(oz - cz);
      const a = dot(dx, dy, dz, dx, dy, dz), hb = dot(ocx, ocy, ocz, dx, dy, dz), cc = f(dot(ocx, ocy, ocz, ocx, ocy, ocz) - f(r * r));
      const disc = f(f(hb * hb) - f(a * cc));
      if (disc < 0) continue;
      const sq = f(Math.sqrt(disc)), t0 = f(f(-hb - sq) / a), t1 = f(f(-hb + sq) / a), t = t0 > EPS_TRI ? t0 : t1;
      if (t > EPS_TRI && t < H.t) {
        H.t = t; H.tri = (0x40000000 | si) >>> 0;
        const qx = f(f(ox + f(dx * t)) - cx), qy = f(f(oy + f(dy * t)) - cy), qz = f(f(oz + f(dz * t)) - cz);
        const il = f(1 / f(Math.sqrt(dot(qx, qy, qz, qx, qy, qz))));
        H.nx = f(qx * il); H.ny = f(qy * il); H.nz = f(qz * il);
        if (anyhit) return true;
      }
    }
    return H.tri !== INVALID;
  }

  traverse(ox, oy, oz, dx, dy, dz, anyhit) {
    if (this.brute) return this.bruteTrace(ox, oy, oz, dx, dy, dz, anyhit);
    const H = this.hit; H.t = INF_T; H.tri = INVALID;
    if (this.numNodes === 0 || this.numTris === 0) return false;
    const ix = Math.abs(dx) > 1e-8 ? f(1 / dx) : INF_T, iy = Math.abs(dy) > 1e-8 ? f(1 / dy) : INF_T, iz = Math.abs(dz) > 1e-8 ? f(1 / dz) : INF_T;
    const b = this.bvh, T = this.tris, st = this.stats, stack = this.stack;
    let sp = 0; stack[0] = 0; st.nodesExamined++;
    const cIdx = [0, 0, 0, 0], cDist = [0, 0, 0, 0];
    while (sp >= 0) {
      const ni = stack[sp--];
      if (isNaN(this.slab(ni, ox, oy, oz, ix, iy, iz, H.t))) continue;
      const base = 1 + ni * 8, meta = b[base + 7];
      if (meta & LEAF) {
        const ti = meta & 0x7fffffff;
        if (ti < this.numTris) {
          st.trisTested++;
          const o = ti * 9;
          const v0x = T[o], v0y = T[o + 1], v0z = T[o + 2];
          const e1x = f(T[o + 3] - v0x), e1y = f(T[o + 4] - v0y), e1z = f(T[o + 5] - v0z);
          const e2x = f(T[o + 6] - v0x), e2y = f(T[o + 7] - v0y), e2z = f(T[o + 8] - v0z);
          const px = f(f(dy * e2z) - f(dz * e2y)), py = f(f(dz * e2x) - f(dx * e2z)), pz = f(f(dx * e2y) - f(dy * e2x));
          const det = dot(e1x, e1y, e1z, px, py, pz);
          if (!(Math.abs(det) < EPS_TRI)) {
            const inv = f(1 / det), sx = f(ox - v0x), sy = f(oy - v0y), sz = f(oz - v0z);
            const u = f(inv * dot(sx, sy, sz, px, py, pz));
            if (!(u < 0 || u > 1)) {
              const qx = f(f(sy * e1z) - f(sz * e1y)), qy = f(f(sz * e1x) - f(sx * e1z)), qz = f(f(sx * e1y) - f(sy * e1x));
              const v = f(inv * dot(dx, dy, dz, qx, qy, qz));
              if (!(v < 0 || f(u + v) > 1)) {
                const t = f(inv * dot(e2x, e2y, e2z, qx, qy, qz));
                if (t > EPS_TRI && t < H.t) {
                  H.t = t; H.tri = ti;
                  const cx = f(f(e1y * e2z) - f(e1z * e2y)), cy = f(f(e1z * e2x) - f(e1x * e2z)), cz = f(f(e1x * e2y) - f(e1y * e2x));
                  const il = f(1 / f(Math.sqrt(dot(cx, cy, cz, cx, cy, cz))));
                  H.nx = f(cx * il); H.ny = f(cy * il); H.nz = f(cz * il);
                  if (anyhit) return true;
                }
              }
            }
          }
        }
        continue;
      }
      let cc = 0;
      for (let c = 0; c < 4; c++) {
        const ci = b[base + 3 + c];
        if (ci === INVALID || ci >= this.numNodes) continue;
        st.nodesExamined++;
        const tm = this.slab(ci, ox, oy, oz, ix, iy, iz, H.t);
        if (!isNaN(tm)) { cIdx[cc] = ci; cDist[cc] = tm; cc++; }
      }
      let best = 0;
      for (let i = 1; i < cc; i++) if (cDist[i] < cDist[best]) best = i;
      if (best !== 0) { const ti = cIdx[0], td = cDist[0]; cIdx[0] = cIdx[best]; cDist[0] = cDist[best]; cIdx[best] = ti; cDist[best] = td; }
      for (let i = cc - 1; i >= 0; i--) if (sp + 1 < STACK_MAX) stack[++sp] = cIdx[i];
    }
    return H.tri !== INVALID;
  }

  primaryRay(P, fx, fy, out) {                               // renderer.wgsl:387-395
    const uvx = f(fx / P.width), uvy = f(fy / P.height);
    const px = fmaf(uvx, 2, -1), py = fmaf(uvy, 2, -1);
    let vx = f(px * P.aspect), vy = py, vz = f(-P.focal);
    const il = f(1 / f(Math.sqrt(dot(vx, vy, vz, vx, vy, vz))));
    vx = f(vx * il); vy = f(vy * il); vz = f(vz * il);
    const q = P.camQuat, ux = q[0], uy = q[1], uz = q[2], s = q[3];   // rotateVectorByQuat, :66-72
    const ax = f(f(uy * vz) - f(uz * vy)), ay = f(f(uz * vx) - f(ux * vz)), az = f(f(ux * vy) - f(uy * vx));
    const bx = f(f(uy * az) - f(uz * ay)), by = f(f(uz * ax) - f(ux * az)), bz = f(f(ux * ay) - f(uy * ax));
    out[0] = fmaf(2, fmaf(s, ax, bx), vx); out[1] = fmaf(2, fmaf(s, ay, by), vy); out[2] = fmaf(2, fmaf(s, az, bz), vz);
  }

  pathSample(P, px, py, sidx, rad) {                         // DESIGN.md section 4
    const key = sampleKey(P.seed, py * P.width + px, sidx), L = this.L, H = this.hit, st = this.stats, d = [0, 0, 0];
    this.primaryRay(P, f(px + rnd(key, 0, 0)), f(py + rnd(key, 0, 1)), d);
    let ox = P.camPos[0], oy = P.camPos[1], oz = P.camPos[2], dx = d[0], dy = d[1], dz = d[2];
    let Tx = 1, Ty = 1, Tz = 1; rad[0] = 0; rad[1] = 0; rad[2] = 0;
    const bR = f(0.9), bG = f(0.7), bB = f(0.3);
    for (let bounce = 0; ; bounce++) {
      st.raysClosest++;
      if (!this.traverse(ox, oy, oz, dx, dy, dz, false)) {
        const e = bounce === 0 ? f(0.01) : f(0.15);
        rad[0] = f(rad[0] + f(Tx * e)); rad[1] = f(rad[1] + f(Ty * e)); rad[2] = f(rad[2] + f(Tz * e));
        return;
      }
      const t = H.t; let nx = H.nx, ny = H.ny, nz = H.nz;
      const hx = f(ox + f(dx * t)), hy = f(oy + f(dy * t)), hz = f(oz + f(dz * t));
      if (!(dot(nx, ny, nz, dx, dy, dz) < 0)) { nx = -nx; ny = -ny; nz = -nz; }
      const sx = f(hx + f(nx * f(1e-4))), sy = f(hy + f(ny * f(1e-4))), sz = f(hz + f(nz * f(1e-4)));
      const ndl = dot(nx, ny, nz, L[0], L[1], L[2]);
      if (ndl > 0) {
        st.raysShadow++;
        if (!this.traverse(sx, sy, sz, L[0], L[1], L[2], true)) {
          rad[0] = f(rad[0] + f(f(Tx * bR) * ndl)); rad[1] = f(rad[1] + f(f(Ty * bG) * ndl)); rad[2] = f(rad[2] + f(f(Tz * bB) * ndl));
        }
      }
      if (bounce >= P.maxBounces) return;
      Tx = f(Tx * bR); Ty = f(Ty * bG); Tz = f(Tz * bB);
      if (bounce >= 2) {
        const p = wmax(wmax(Tx, Ty), Tz);
        if (rnd(key, bounce, 4) >= p) return;
        const ip = f(1 / p); Tx = f(Tx * ip); Ty = f(Ty * ip); Tz = f(Tz * ip);
      }
      const u1 = rnd(key, bounce, 2), u2 = rnd(key, bounce, 3), sc = this._sc;
      sincos2pi(u2, sc);
      const r = f(Math.sqrt(u1)), lx = f(r * sc[0]), ly = f(r * sc[1]), lz = f(Math.sqrt(f(1 - u1)));
      const sign = (nz < 0 || Object.is(nz, -0)) ? -1 : 1, a = f(-1 / f(sign + nz)), b = f(f(nx * ny) * a);
      const tx = f(1 + f(f(f(sign * nx) * nx) * a)), ty = f(sign * b), tz = f(f(-sign) * nx);
      const ux = b, uy = f(sign + f(f(ny * ny) * a)), uz = -ny;
      dx = f(f(f(tx * lx) + f(ux * ly)) + f(nx * lz)); dy = f(f(f(ty * lx) + f(uy * ly)) + f(ny * lz)); dz = f(f(f(tz * lx) + f(uz * ly)) + f(nz * lz));
      ox = sx; oy = sy; oz = sz;
    }
  }

  // P: {width,height,focal,aspect,camPos[3],camQuat[4],frame,mode(1|2),spp,maxBounces,seed,stepX,stepY}
  render(P, out) {
    const rad = [0, 0, 0], d = [0, 0, 0], L = this.L, H = this.hit, st = this.stats;
    const sx = P.stepX || 1, sy = P.stepY || 1;
    for (let py = 0; py < P.height; py += sy) for (let px = 0; px < P.width; px += sx) {
      const o = (py * P.width + px) * 4;
      if (P.mode === 1) {                                   // reference shading, renderer.wgsl:348-353,410
        this.primaryRay(P, f(px + 0.5), f(py + 0.5), d); st.raysClosest++; st.samples++;
        if (this.traverse(P.camPos[0], P.camPos[1], P.camPos[2], d[0], d[1], d[2], false)) {
          const k = f(f(0.15) + wmax(dot(H.nx, H.ny, H.nz, L[0], L[1], L[2]), 0));
          out[o] = f(f(0.9) * k); out[o + 1] = f(f(0.7) * k); out[o + 2] = f(f(0.3) * k);
        } else { out[o] = out[o + 1] = out[o + 2] = f(0.01); }
      } else {
        let r = 0, g = 0, b = 0;
        for (let s = 0; s < P.spp; s++) { this.pathSample(P, px, py, P.frame * P.spp + s, rad); r = f(r + rad[0]); g = f(g + rad[1]); b = f(b + rad[2]); st.samples++; }
        const inv = f(1 / P.spp); out[o] = f(r * inv); out[o + 1] = f(g * inv); out[o + 2] = f(b * inv);
      }
      out[o + 3] = 1;
    }
    return st;
  }
}

module.exports = { Oracle, fmaf, halfToFloat, rnd, sampleKey };

// CLI: node oracle/js/pt_oracle.js <tris.f32> <bvh4.u32> <params.json> <out.f32>  -> prints {"seconds":..,"stats":..}
if (require.main === module) {
  const fs = require("fs"), a = process.argv;
  const rd = (p, T) => { const b = fs.readFileSync(p); return new T(b.buffer.slice(b.byteOffset, b.byteOffset + b.byteLength)); };
  const tris = rd(a[2], Float32Array), bvh = rd(a[3], Uint32Array), P = JSON.parse(fs.readFileSync(a[4], "utf8"));
  const spheres = P.spheres ? new Float32Array(P.spheres) : null;
  P.focal = f(P.focal); P.aspect = f(P.aspect); P.camPos = P.camPos.map(f); P.camQuat = P.camQuat.map(f);
  const out = new Float32Array(P.width * P.height * 4), orc = new Oracle(tris, bvh, P.numTris, spheres);
  const t0 = Date.now(), st = orc.render(P, out), sec = (Date.now() - t0) / 1000;
  fs.writeFileSync(a[5], Buffer.from(out.buffer));
  console.log(JSON.stringify({ seconds: sec, stats: st, node: process.version }));
}
