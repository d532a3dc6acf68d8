// ThreadSanitizer harness for the record emitter (twk_record_sink.h): `make tsan` builds the host sources with
// -fsanitize=thread around this file and runs it.  Streams sorted survivors in ragged pieces through RecordEmitter with
// 1 / 5 / 16 workers, with and without a backlog of expanded blocks, the mapped output and the hand-off queue (RecordHandOff), reads the files back and
// compares record counts; any data race TSan sees fails the run.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unistd.h>
#include <vector>

#include "twk_format.h"
#include "twk_hip.h"
#include "twk_record_sink.h"

using namespace tomahawk;

int main() {
	const uint32_t M = 3000;
	std::vector<uint32_t> rid(M), pos(M);
	for (uint32_t v = 0; v < M; ++v) { rid[v] = v / 1000; pos[v] = 1000 + 10 * (v % 1000); }
	std::vector<twk_hip_record> recs;
	uint64_t x = 88172645463325252ull;
	for (uint32_t a = 0; a + 1 < M; ++a)
		for (uint32_t b = a + 1; b < std::min(M, a + 1 + 40); ++b) {
			x ^= x << 13; x ^= x >> 7; x ^= x << 17;
			if (x % 3 == 0) continue;
			twk_hip_record r{};
			r.idxA = a; r.idxB = b; r.flags = 3; r.R2 = (double)(x % 1000) / 1000.0; r.D = 0.01; r.cnt[0] = (double)(x % 97);
			recs.push_back(r);
		}
	int bad = 0;
	for (int workers : {1, 5, 16}) for (size_t backlog : {(size_t)0, (size_t)8 << 20}) for (int mapped : {0, 1}) for (size_t queue : {(size_t)0, (size_t)2}) {
		const std::string path = "/tmp/emitter_tsan_" + std::to_string(getpid()) + ".two";
		TwoOutput out;
		Header hdr;
		hdr.literals = "##fileformat=VCFv4.2\n";
		for (int s = 0; s < 4; ++s) hdr.samples.push_back("S" + std::to_string(s));
		for (uint32_t c = 0; c < 3; ++c) { Contig k; k.idx = c; k.name = std::to_string(c + 1); k.n_bases = 250000000; hdr.contigs.push_back(k); }
		if (!out.writer.open(path, hdr, 1)) { fprintf(stderr, "open failed\n"); return 1; }
		if (mapped) (void)out.writer.map_output();
		out.b_size = 700; out.c_level = 1; out.rid = rid.data(); out.pos = pos.data(); out.n_variants = M;
		bool ok = true;
		{
			RecordEmitter em(out, workers, backlog);
			RecordHandOff hand(em, queue);           // (destroyed before the emitter)
			size_t k = 0;
			const size_t steps[] = {1, 699, 700, 701, 5000, 0, 33333, 7};
			for (size_t i = 0; k < recs.size(); ++i) {
				const size_t n = std::min(steps[i % 8], recs.size() - k);
				if (hand.takes(n)) ok = ((i & 1) ? hand.put(recs.data() + k, n, [](twk_hip_record&) { return true; }) : hand.put(recs.data() + k, n)) && ok;
				else ok = hand.drain() && em.emit(recs.data() + k, n, false, true) && ok;
				k += n;
			}
			ok = hand.drain() && ok;
			ok = em.emit(nullptr, 0, true) && ok;
		}
		ok = out.writer.close() && ok;
		if (!ok || out.n_records != 2 * recs.size()) { fprintf(stderr, "workers %d backlog %zu mapped %d queue %zu: wrote %llu of %zu\n", workers, backlog, mapped, queue, (unsigned long long)out.n_records, 2 * recs.size()); ++bad; }
		unlink(path.c_str());
	}
	printf("emitter_tsan: %zu survivors x 24 configurations, %d bad\n", recs.size(), bad);
	return bad ? 1 : 0;
}
