#!/usr/bin/env node
// Writes tests/golden/unicode14_lower_additions.json: the simple-lowercase pairs node/ICU (Unicode 14.0) has
// beyond what this image's Python (Unicode 13.0) knows, with their character names where node can tell... it cannot,
// so only code points.  Data, not code: the test suite reads it so that CPU tests do not need node at run time.
'use strict';
const fs = require('fs'), path = require('path'), cp = require('child_process');
const py = "import unicodedata,json;print(json.dumps([c for c in range(0x110000) if not (0xD800<=c<=0xDFFF) and unicodedata.category(chr(c))=='Cn']))";
const unassigned13 = new Set(JSON.parse(cp.execFileSync('python3', ['-c', py], {maxBuffer: 1 << 28}).toString()));
const pairs = [];
for (let c = 0; c < 0x110000; c++) {
  if (c >= 0xD800 && c <= 0xDFFF) continue;
  let l = Array.from(String.fromCodePoint(c).toLowerCase());
  if (c === 0x130) l = ['i'];
  const t = l[0].codePointAt(0);
  if (l.length === 1 && t !== c && unassigned13.has(c)) pairs.push([c, t]);
}
const out = {source: 'node ' + process.versions.node + ' / ICU ' + process.versions.icu + ' / Unicode ' + process.versions.unicode,
  note: 'simple lowercase pairs (from, to) of code points unassigned in Unicode 13.0', pairs: pairs};
fs.writeFileSync(path.join(__dirname, 'unicode14_lower_additions.json'), JSON.stringify(out) + '\n');
console.log(pairs.length, 'pairs');
