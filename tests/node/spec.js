// Drives bindings/node (the reference's DasContextJs surface over libc_eth_kzg.so) on one golden vector:
//   node spec.js <blob file> <expected: 128*2048 cell bytes | 128*48 proof bytes | 48 commitment bytes>
// Mirrors bindings/node/__test__/index.spec.ts of the reference: sync and async forms, verification, recovery, errors.
'use strict'
const fs = require('fs')
const path = require('path')
const kzg = require(path.join(__dirname, '..', '..', 'bindings', 'node'))

const eq = (a, b) => Buffer.compare(Buffer.from(a), Buffer.from(b)) === 0
async function main () {
  const blob = new Uint8Array(fs.readFileSync(process.argv[2]))
  const want = fs.readFileSync(process.argv[3])
  const wantCells = [], wantProofs = []
  for (let i = 0; i < 128; i++) wantCells.push(want.subarray(i * 2048, (i + 1) * 2048))
  for (let i = 0; i < 128; i++) wantProofs.push(want.subarray(128 * 2048 + i * 48, 128 * 2048 + (i + 1) * 48))
  const wantCommitment = want.subarray(128 * 2048 + 128 * 48)
  const out = {}
  const ctx = kzg.DasContextJs.create({ usePrecomp: true })
  out.constants = kzg.BYTES_PER_BLOB === 131072 && kzg.BYTES_PER_CELL === 2048 && kzg.MAX_NUM_COLUMNS === 128 && kzg.BYTES_PER_PROOF === 48

  const cp = ctx.computeCellsAndKzgProofs(blob)
  out.sync_cells = cp.cells.length === 128 && cp.cells.every((c, i) => eq(c, wantCells[i]))
  out.sync_proofs = cp.proofs.length === 128 && cp.proofs.every((p, i) => eq(p, wantProofs[i]))
  out.sync_commitment = eq(ctx.blobToKzgCommitment(blob), wantCommitment)
  out.sync_cells_only = ctx.computeCells(blob).every((c, i) => eq(c, wantCells[i]))

  // async forms: eight calls in flight on ONE context (libuv worker threads), as the reference's async_* methods allow
  const jobs = []
  for (let k = 0; k < 4; k++) { jobs.push(ctx.asyncComputeCellsAndKzgProofs(blob)); jobs.push(ctx.asyncBlobToKzgCommitment(blob)) }
  const res = await Promise.all(jobs)
  out.async_all = res.every((r, k) => k % 2 ? eq(r, wantCommitment) : (r.cells.every((c, i) => eq(c, wantCells[i])) && r.proofs.every((p, i) => eq(p, wantProofs[i]))))

  const commitments = cp.cells.map(() => wantCommitment)
  const idx = cp.cells.map((_, i) => (i % 2 ? BigInt(i) : i))  // number | bigint, as index.d.ts allows
  out.verify_true = ctx.verifyCellKzgProofBatch(commitments, idx, cp.cells, cp.proofs) === true
  const badProofs = cp.proofs.slice(); badProofs[5] = cp.proofs[6]
  out.verify_false = (await ctx.asyncVerifyCellKzgProofBatch(commitments, idx, cp.cells, badProofs)) === false
  out.verify_empty = ctx.verifyCellKzgProofBatch([], [], [], []) === true

  const half = [], halfIdx = []
  for (let i = 1; i < 128; i += 2) { half.push(cp.cells[i]); halfIdx.push(i) }
  const rec = await ctx.asyncRecoverCellsAndKzgProofs(halfIdx, half)
  out.recover = rec.cells.every((c, i) => eq(c, wantCells[i])) && rec.proofs.every((p, i) => eq(p, wantProofs[i]))

  // errors are exceptions / rejections with the reference's message shape
  const bad = new Uint8Array(131072).fill(0xff)
  try { ctx.computeCellsAndKzgProofs(bad); out.error_sync = false } catch (e) { out.error_sync = /failed to compute compute_cells_and_kzg_proofs/.test(e.message) }
  try { await ctx.asyncComputeCells(bad); out.error_async = false } catch (e) { out.error_async = /failed to compute compute_cells/.test(e.message) }
  try { ctx.blobToKzgCommitment(new Uint8Array(5)); out.error_length = false } catch (e) { out.error_length = /blob must be a Uint8Array/.test(e.message) }
  try { ctx.recoverCellsAndKzgProofs(halfIdx.slice(0, 10), half.slice(0, 10)); out.error_recover = false } catch (e) { out.error_recover = true }

  // EIP-4844 single-point operations
  const z = new Uint8Array(32); z[31] = 9
  const [proof, y] = ctx.computeKzgProof(blob, z)
  out.kzg_proof = proof.length === 48 && y.length === 32 && ctx.verifyKzgProof(wantCommitment, z, y, proof) === true
  const y2 = y.slice(); y2[31] ^= 1
  out.kzg_proof_false = (await ctx.asyncVerifyKzgProof(wantCommitment, z, y2, proof)) === false
  const blobProof = await ctx.asyncComputeBlobKzgProof(blob, wantCommitment)
  out.blob_proof = ctx.verifyBlobKzgProof(blob, wantCommitment, blobProof) === true
  out.blob_proof_batch = ctx.verifyBlobKzgProofBatch([blob, blob], [wantCommitment, wantCommitment], [blobProof, blobProof]) === true &&
    (await ctx.asyncVerifyBlobKzgProofBatch([blob], [wantCommitment], [proof])) === false

  // a second context (the reference's tests create several): shares the window tables
  const ctx2 = new kzg.DasContextJs()
  out.second_context = eq(ctx2.blobToKzgCommitment(blob), wantCommitment)
  console.log(JSON.stringify(out))
}
main().catch((e) => { console.error(e); process.exit(1) })
