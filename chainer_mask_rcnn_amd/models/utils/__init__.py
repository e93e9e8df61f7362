# flake8: noqa
from .proposal_target_creator import ProposalTargetCreator
from .anchor_target_creator import AnchorTargetCreator
from .proposal_creator import ProposalCreator
