"""Per-kernel means of the rocprofv3 --pmc passes written by tools/profile_bench.sh -> JSON on stdout."""
import csv, glob, json, os, sys
from collections import defaultdict


def short(name):
    for key in ('neutra_grad_wide_kernel', 'neutra_leapfrog_wide_kernel', 'realnvp_forward_wide_kernel', 'realnvp_inverse_wide_kernel', 'flow_mh_wide_kernel',
                'fit_rows_kernel', 'fit_grad_kernel', 'fit_fold_kernel', 'rows_sample_kernel', 'blob_copy_kernel', 'mala_kernel', 'flow_mh_b2_kernel', 'flow_mh_b_kernel', 'stats_finish_kernel', 'tune_finish_kernel', 'neutra_hmc_kernel', 'hmc_kernel',
                'imh_eval_kernel', 'imh_scan_kernel', 'imh_replay_kernel', 'neutra_leapfrog_mfma_kernel',
                'neutra_grad_mfma_kernel', 'flow_mh_mfma_kernel', 'flow_mh_kernel', 'realnvp_forward', 'realnvp_inverse'):
        if key in name:
            # the opt-in Philox4x32-7 instantiations (last template argument 7) are different kernels
            return key + '_philox7' if (', 7>(' in name and key in ('mala_kernel', 'hmc_kernel', 'flow_mh_b_kernel', 'flow_mh_b2_kernel')) else key
    return None


def main(root):
    acc = defaultdict(lambda: defaultdict(list))
    meta = {}
    for path in glob.glob(os.path.join(root, 'pmc_*', '**', '*counter_collection.csv'), recursive=True):
        with open(path) as fh:
            for row in csv.DictReader(fh):
                k = short(row.get('Kernel_Name', ''))
                if k is None:
                    continue
                acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
                meta.setdefault(k, {'kernel': row['Kernel_Name'][:110], 'vgpr': row.get('VGPR_Count'),
                                    'sgpr': row.get('SGPR_Count'), 'lds': row.get('LDS_Block_Size'),
                                    'grid': row.get('Grid_Size'), 'wg': row.get('Workgroup_Size')})
    out = {}
    try:   # the digest of the library that was profiled (the tree's in-tree build): bench.py flags a summary of other code
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from nfmc_amd import hip
        out['_meta'] = {'library_digest': hip.build_digest(), 'source': root}
    except Exception as e:   # noqa
        out['_meta'] = {'library_digest': None, 'error': repr(e)}
    for k, counters in acc.items():
        out[k] = dict(meta[k])
        for c, vals in sorted(counters.items()):
            out[k][c] = sum(vals) / len(vals)
        out[k]['launches_sampled'] = max(len(v) for v in counters.values())
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/prof')
