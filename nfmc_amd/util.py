"""Flow-spec strings and the Metropolis log-ratio (nfmc/util.py:189-215, 218-281, 382-392).

The half-split coupling flows `realnvp`, `nice` and `c-rqnsf` are on this build's path; the reference's other ~57
architecture names live in torchflows and raise a clear error here instead of silently mapping to something else.
"""
import json
from typing import Dict, List

FLOW_NAMES: Dict[str, List[str]] = {
    'realnvp': ['realnvp', 'real_nvp', 'rnvp'],  # nfmc/util.py:6
    'nice': ['nice'],                            # nfmc/util.py:13
    'c-rqnsf': ['c-rqnsf', 'c-rqsnsf'],          # nfmc/util.py:17
}


def is_flow_supported(flow_name: str):
    return any(flow_name in v for v in FLOW_NAMES.values())


def get_supported_normalizing_flows(synonyms: bool = True):
    if synonyms:
        return sorted({n for v in FLOW_NAMES.values() for n in v})
    return sorted(FLOW_NAMES.keys())


def parse_flow_string(flow_string: str):
    """`<flow_name>%<json>` or `<flow_name>` (nfmc/util.py:189-215)."""
    if flow_string is None:
        return {'name': None, 'kwargs': {}, 'hash': hash('None')}
    if '%' not in flow_string:
        return {'name': flow_string, 'kwargs': {}, 'hash': hash(flow_string)}
    flow_name = flow_string.split('%')[0]
    kwargs = json.loads(flow_string.split('%')[1])
    return {'name': flow_name, 'kwargs': kwargs, 'hash': hash(flow_name + str(kwargs))}


def create_flow_object(flow_string: str, event_shape, **kwargs):
    """nfmc/util.py:218-281,379 for the realnvp branch."""
    from .flows import CRQNSF, NICE, Flow, RealNVP
    data = parse_flow_string(flow_string)
    name = data['name']
    kwargs.update(data['kwargs'])
    if not isinstance(name, str):
        raise ValueError
    if not is_flow_supported(name):
        raise ValueError(f"flow '{name}' is outside this build's path (supported: {get_supported_normalizing_flows()})")
    cls = NICE if name in FLOW_NAMES['nice'] else (CRQNSF if name in FLOW_NAMES['c-rqnsf'] else RealNVP)
    return Flow(cls(event_shape, **kwargs))


def metropolis_acceptance_log_ratio(log_prob_target_curr, log_prob_target_prime, log_prob_proposal_curr,
                                    log_prob_proposal_prime):
    """log [ p(x') g(x|x') / (p(x) g(x'|x)) ]  (nfmc/util.py:382-392)."""
    return log_prob_target_prime - log_prob_target_curr + log_prob_proposal_curr - log_prob_proposal_prime


def get_supported_mcmc_samplers() -> List[str]:
    return ['hmc', 'uhmc', 'ula', 'mala', 'mh']


def get_supported_nfmc_samplers() -> List[str]:
    return ['imh', 'fixed_imh', 'adaptive_imh', 'jump_mala', 'jump_ula', 'jump_hmc', 'jump_uhmc', 'jump_mh', 'neutra_hmc', 'neutra_mh']


def get_supported_samplers() -> List[str]:
    return get_supported_mcmc_samplers() + get_supported_nfmc_samplers()
