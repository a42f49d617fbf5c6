/*
 * megalania-hip -- command-line driver in C, the GPU-path counterpart of the reference's
 * main.c:28-128: `megalania-hip [options] <file>` writes an LZMA-alone stream to stdout and
 * progress to stderr.  Host code only calls the C ABI (include/megalania_hip.h) and emits
 * the best slab through EncoderInterface / OutputInterface like main.c:110-119.
 *
 * Schedule (main.c:64-77): `phases` x `epochs` epochs; phase 0 epochs start from an
 * all-literal slab, later phases from the best slab so far.  The reference runs N (= file
 * size) single-neighbour iterations per epoch; here an epoch is ceil(N / K) steps of K
 * neighbours, i.e. the same number of neighbour evaluations.  Defaults are the reference's
 * (3 x 200) and, like the reference, take a very long time on anything but tiny inputs:
 * use --epochs / --steps to bound the run.
 */
#include <fcntl.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include "mgl_host.h"

static void usage(const char* argv0)
{
	fprintf(stderr,
	        "usage: %s [--neighbours K] [--epochs E] [--phases P] [--steps S] [--seed N]\n"
	        "          [--lc N --lp N --pb N] [--device D] [--max-scan M]\n"
	        "          [-o out.lzma] [--save-slab file] [--load-slab file] [--greedy-seed C] [--temperature B]\n"
	        "          [--accept auto|single|bulk] [--chains N --rank R --comm-file PATH [--comm-nonce X] [--transport rccl|shm]] filename\n"
	        "  -o           write the stream to a file instead of stdout\n"
	        "  --save-slab  after every epoch, write the best packet slab (resumable checkpoint)\n"
	        "  --load-slab  start from a slab written by --save-slab (same input, same lc/lp/pb)\n"
	        "  --greedy-seed C  epochs that the reference starts from the all-literal slab start from a greedy\n"
	        "               parse instead (longest of the C nearest candidates per position; e.g. 256)\n"
	        "  --temperature B  Metropolis accept rule instead of the reference's: B = e-folding slack in output\n"
	        "               bytes at the start of an epoch, cooled linearly to 0 (e.g. 2; 0 = reference rule)\n"
	        "  --chains N --rank R --comm-file PATH  one of N independent chains, one process per GPU (device = R unless\n"
	        "               --device says otherwise): after every epoch the chains exchange their best slab over RCCL\n"
	        "               (8-byte all-reduce + one broadcast); rank 0 creates PATH (the communicator id) and writes the\n"
	        "               stream, the others wait for a PATH that carries the same --comm-nonce (give every run its own,\n"
	        "               e.g. the launcher's pid: a file left by an earlier run is then never mistaken for this one's);\n"
	        "               --transport shm stages the exchange through PATH itself (host shared memory) instead of RCCL:\n"
	        "               for chains that share one GPU.  --save-slab: chain R > 0 writes to <file>.rankR\n"
	        "  --accept     what a step of K neighbours takes: the best acceptable one (single), every one that is\n"
	        "               the best of its own window (bulk), or whichever pays (auto, default)\n", argv0);
}

int main(int argc, char** argv)
{
	mgl_sa_config cfg;
	memset(&cfg, 0, sizeof cfg);
	cfg.seed = 1673551; /* main.c:68 */
	cfg.neighbours_per_step = 4096;
	cfg.top_k = 20;     /* main.c:49 */
	mgl_properties props = { 0, 0, 0 }; /* main.c:45 */
	unsigned epochs = 200, phases = 3;  /* main.c:66,69 */
	unsigned long long steps_override = 0;
	const char* filename = NULL;
	const char *out_path = NULL, *save_path = NULL, *load_path = NULL;
	uint32_t greedy = 0;
	double temperature_bytes = 0;
	int accept_mode = MGL_ACCEPT_AUTO;
	int chains = 1, rank = 0, device_given = 0;
	const char* comm_path = NULL;
	unsigned long long comm_nonce = 0;
	int transport_shm = 0;
	for (int i = 1; i < argc; i++) {
		const char* a = argv[i];
		const char* v = i + 1 < argc ? argv[i + 1] : NULL;
		if (a[0] != '-') { filename = a; continue; }
		if (!v) { usage(argv[0]); return -1; }
		if (!strcmp(a, "--neighbours")) cfg.neighbours_per_step = (uint32_t)strtoul(v, NULL, 0);
		else if (!strcmp(a, "--epochs")) epochs = (unsigned)strtoul(v, NULL, 0);
		else if (!strcmp(a, "--phases")) phases = (unsigned)strtoul(v, NULL, 0);
		else if (!strcmp(a, "--steps")) steps_override = strtoull(v, NULL, 0);
		else if (!strcmp(a, "--seed")) cfg.seed = strtoull(v, NULL, 0);
		else if (!strcmp(a, "--lc")) props.lc = (uint8_t)strtoul(v, NULL, 0);
		else if (!strcmp(a, "--lp")) props.lp = (uint8_t)strtoul(v, NULL, 0);
		else if (!strcmp(a, "--pb")) props.pb = (uint8_t)strtoul(v, NULL, 0);
		else if (!strcmp(a, "--device")) { cfg.device = (int32_t)strtol(v, NULL, 0); device_given = 1; }
		else if (!strcmp(a, "--chains")) chains = (int)strtol(v, NULL, 0);
		else if (!strcmp(a, "--rank")) rank = (int)strtol(v, NULL, 0);
		else if (!strcmp(a, "--comm-file")) comm_path = v;
		else if (!strcmp(a, "--comm-nonce")) comm_nonce = strtoull(v, NULL, 0);
		else if (!strcmp(a, "--transport")) {
			if (!strcmp(v, "shm")) transport_shm = 1;
			else if (strcmp(v, "rccl") != 0) { usage(argv[0]); return -1; }
		}
		else if (!strcmp(a, "--max-scan")) cfg.max_bucket_scan = (uint32_t)strtoul(v, NULL, 0);
		else if (!strcmp(a, "-o")) out_path = v;
		else if (!strcmp(a, "--save-slab")) save_path = v;
		else if (!strcmp(a, "--load-slab")) load_path = v;
		else if (!strcmp(a, "--greedy-seed")) greedy = (uint32_t)strtoul(v, NULL, 0);
		else if (!strcmp(a, "--temperature")) temperature_bytes = strtod(v, NULL);
		else if (!strcmp(a, "--accept")) {
			if (!strcmp(v, "auto")) accept_mode = MGL_ACCEPT_AUTO;
			else if (!strcmp(v, "single")) accept_mode = MGL_ACCEPT_SINGLE;
			else if (!strcmp(v, "bulk")) accept_mode = MGL_ACCEPT_BULK;
			else { usage(argv[0]); return -1; }
		}
		else { usage(argv[0]); return -1; }
		i++;
	}
	if (!filename) { usage(argv[0]); return -1; }
	if (chains < 1 || rank < 0 || rank >= chains || (chains > 1 && !comm_path)) { usage(argv[0]); return -1; }
	if (chains > 1) {
		if (!device_given) cfg.device = rank;
		/* distinct, reproducible RNG stream per chain (the same rule as megalania_amd/multi_gpu.py:chain_seed) */
		if (rank) cfg.seed ^= 0x9E3779B97F4A7C15ull * (uint64_t)(rank + 1);
	}

	int fd = open(filename, O_RDONLY);
	if (fd < 0) { fprintf(stderr, "Error: could not open %s\n", filename); return -1; }
	struct stat sb;
	if (fstat(fd, &sb) < 0) { fprintf(stderr, "Error: could not stat %s\n", filename); close(fd); return -1; }
	const size_t file_size = (size_t)sb.st_size;
	if (file_size == 0) { close(fd); return 0; } /* main.c:40-42 */
	const uint8_t* file_data = (const uint8_t*)mmap(NULL, file_size, PROT_READ, MAP_PRIVATE, fd, 0);
	if (file_data == MAP_FAILED) { fprintf(stderr, "Error: could not map %s\n", filename); close(fd); return -1; }

	if (mgl_device_count() < 1) {
		fprintf(stderr, "Error: no HIP device found; this program has no CPU search path\n");
		return -1;
	}
	cfg.iters_per_epoch = file_size;
	mgl_sa* sa = mgl_sa_create(file_data, file_size, props, &cfg);
	if (sa == NULL) { fprintf(stderr, "Error: %s\n", mgl_last_error()); return -1; }

	if (temperature_bytes > 0 && mgl_sa_set_temperature(sa, (uint64_t)(temperature_bytes * 16384.0)) != MGL_OK) {
		fprintf(stderr, "Error: %s\n", mgl_last_error());
		return -1;
	}
	mgl_comm* comm = NULL;
	if (comm_path && transport_shm) {
		if (mgl_comm_init_shm(&comm, comm_path, comm_nonce, rank, chains, cfg.device) != MGL_OK) { fprintf(stderr, "Error: %s\n", mgl_last_error()); return -1; }
	} else if (comm_path) { /* also with --chains 1: the same code path on a one-GPU box */
		/* rendezvous file: "MGLCOMM1", u64 nonce, the 128-byte RCCL id.  Rank 0 replaces whatever an earlier run left
		 * under the name and removes it again once every chain has joined; the others only take a file of this run */
		uint8_t uid[128];
		if (rank == 0) {
			char tmp[4096];
			snprintf(tmp, sizeof tmp, "%s.tmp", comm_path);
			(void)unlink(comm_path);
			FILE* f = fopen(tmp, "wb");
			const uint64_t nonce = comm_nonce;
			if (mgl_comm_unique_id(uid) != MGL_OK || !f || fwrite("MGLCOMM1", 8, 1, f) != 1 || fwrite(&nonce, 8, 1, f) != 1 ||
			    fwrite(uid, 128, 1, f) != 1 || fclose(f) != 0 || rename(tmp, comm_path) != 0) {
				fprintf(stderr, "Error: could not publish the communicator id in %s: %s\n", comm_path, mgl_last_error());
				return -1;
			}
		} else {
			const char* te = getenv("MGL_COMM_TIMEOUT_S");
			const double limit = te && atof(te) > 0 ? atof(te) : 600.0;
			bool got = false;
			for (double waited = 0; !got && waited < limit; waited += 0.1) {
				FILE* f = fopen(comm_path, "rb");
				char magic[8];
				uint64_t nonce = 0;
				if (f && fread(magic, 8, 1, f) == 1 && !memcmp(magic, "MGLCOMM1", 8) && fread(&nonce, 8, 1, f) == 1 && nonce == comm_nonce &&
				    fread(uid, 128, 1, f) == 1) got = true;
				if (f) fclose(f);
				if (!got) usleep(100000);
			}
			if (!got) { fprintf(stderr, "Error: no communicator id of this run (--comm-nonce %llu) appeared in %s\n", comm_nonce, comm_path); return -1; }
		}
		if (mgl_comm_init(&comm, uid, rank, chains, cfg.device) != MGL_OK) { fprintf(stderr, "Error: %s\n", mgl_last_error()); return -1; }
		if (rank == 0) (void)unlink(comm_path); /* ncclCommInitRank returns once every rank has joined: nobody reads the file any more */
	}
	if (mgl_sa_set_accept_mode(sa, accept_mode, 0) != MGL_OK) { fprintf(stderr, "Error: %s\n", mgl_last_error()); return -1; }
	mgl_packet* packets_best = (mgl_packet*)malloc(sizeof(mgl_packet) * file_size);
	if (packets_best == NULL) { fprintf(stderr, "Error: out of memory\n"); return -1; }
	bool resumed = false;
	if (load_path) {
		/* slab file: "MGLSLAB1", u64 size, u64 perplexity, size x 12-byte packets (lzma_packet.h:13-17) */
		FILE* f = fopen(load_path, "rb");
		char magic[8];
		uint64_t hdr[2] = { 0, 0 };
		if (!f || fread(magic, 8, 1, f) != 1 || memcmp(magic, "MGLSLAB1", 8) != 0 || fread(hdr, 8, 2, f) != 2 ||
		    hdr[0] != file_size || fread(packets_best, sizeof(mgl_packet), file_size, f) != file_size) {
			fprintf(stderr, "Error: %s is not a slab for this input\n", load_path);
			return -1;
		}
		fclose(f);
		/* the library re-costs the slab and refuses it unless the perplexity matches */
		if (mgl_sa_set_best(sa, packets_best, hdr[1]) != MGL_OK) { fprintf(stderr, "Error: %s\n", mgl_last_error()); return -1; }
		resumed = true;
	}

	unsigned long long steps_per_epoch = (file_size + cfg.neighbours_per_step - 1) / cfg.neighbours_per_step;
	if (steps_override) steps_per_epoch = steps_override;
	for (unsigned phase = 0; phase < phases; phase++) {
		for (unsigned epoch = 0; epoch < epochs; epoch++) {
			if (mgl_sa_begin_epoch(sa, phase, phase != 0 || resumed) != MGL_OK) { fprintf(stderr, "Error: %s\n", mgl_last_error()); return -1; }
			if (greedy && phase == 0 && !resumed && mgl_sa_seed_greedy(sa, greedy) != MGL_OK) { fprintf(stderr, "Error: %s\n", mgl_last_error()); return -1; }
			mgl_sa_stats st;
			if (mgl_sa_run(sa, steps_per_epoch, &st) != MGL_OK) { fprintf(stderr, "Error: %s\n", mgl_last_error()); return -1; }
			/* main.c:97-99: 18 = 13 header bytes + 5 flush bytes, 16384 = 2048 * 8 */
			fprintf(stderr, "current file size: %f\tbest: %f\tstep: %u\tepoch: %04u\t%.0f evals/s\n",
			        18 + st.current_cost / 16384.f, 18 + st.best_cost / 16384.f, phase + 1, epoch,
			        st.gpu_ms_total > 0 ? st.evaluations / (st.gpu_ms_total * 1e-3) : 0.0);
			if (comm) {
				int winner = -1;
				uint64_t wcost = 0;
				if (mgl_sa_exchange_best(sa, comm, &winner, &wcost) != MGL_OK) { fprintf(stderr, "Error: %s\n", mgl_last_error()); return -1; }
				fprintf(stderr, "exchange: chain %d holds the best slab, %f bytes\n", winner, 18 + wcost / 16384.f);
			}
			if (save_path && st.best_cost != 0) {
				uint64_t hdr[2] = { file_size, 0 };
				if (mgl_sa_best(sa, packets_best, &hdr[1]) != MGL_OK) { fprintf(stderr, "Error: %s\n", mgl_last_error()); return -1; }
				/* one checkpoint per chain: chain R > 0 writes <file>.rankR (the chains run the same command line) */
				char dst[4096], tmp[4200];
				if (rank) snprintf(dst, sizeof dst, "%s.rank%d", save_path, rank);
				else snprintf(dst, sizeof dst, "%s", save_path);
				snprintf(tmp, sizeof tmp, "%s.tmp", dst);
				FILE* f = fopen(tmp, "wb");
				if (!f || fwrite("MGLSLAB1", 8, 1, f) != 1 || fwrite(hdr, 8, 2, f) != 2 ||
				    fwrite(packets_best, sizeof(mgl_packet), file_size, f) != file_size || fclose(f) != 0 || rename(tmp, dst) != 0)
					fprintf(stderr, "warning: could not write %s\n", dst);
			}
		}
	}

	uint64_t best = 0;
	if (mgl_sa_best(sa, packets_best, &best) != MGL_OK) {
		fprintf(stderr, "Error: could not fetch the best slab: %s\n", mgl_last_error());
		return -1;
	}
	mgl_comm_destroy(comm);
	mgl_sa_destroy(sa);
	if (rank != 0) return 0; /* every chain holds the common best slab after the last exchange; rank 0 writes it */

	FILE* out = stdout;
	if (out_path && (out = fopen(out_path, "wb")) == NULL) { fprintf(stderr, "Error: could not open %s\n", out_path); return -1; }
	OutputInterface output;
	mgl_file_output_new(&output, out);
	if (!mgl_emit_stream(file_data, file_size, props, packets_best, &output)) return -1;
	if (out_path ? fclose(out) != 0 : fflush(stdout) != 0) { fprintf(stderr, "Error: could not write the stream\n"); return -1; }
	free(packets_best);
	munmap((void*)file_data, file_size);
	close(fd);
	return 0;
}
